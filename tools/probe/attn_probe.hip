// Timing variants of the dh = 40 attention kernels on the step's own shape (B16 H8 T = S = 4096, fused q|k|v rows): which of
// MFMA / softmax VALU / v_exp_f32 / tile staging the time goes to, and how far they overlap.  Built once per variant:
//   for v in 0 1 2 3 4; do hipcc --offload-arch=gfx950 -O3 -std=c++17 -DATTN_PROBE=$v tools/probe/attn_probe.hip \
//       e4t-diffusion_amd/csrc/obj/core.o -o /tmp/attn_probe_$v; done
// (variant 0 is the product kernel; the others compute wrong results by construction — see attention.hip, ATTN_PROBE)
#include "../../e4t-diffusion_amd/csrc/attention.hip"
#include <cstdio>
#include <vector>

int main() {
  const int B = 16, H = 8, T = 4096, DH = 40, d = H * DH, ld = 3 * d;
  const size_t n = (size_t)B * T * ld;
  std::vector<unsigned short> h(n);
  unsigned x = 12345u;
  for (size_t i = 0; i < n; ++i) { x = x * 1664525u + 1013904223u; h[i] = (unsigned short)(0x3C00u + ((x >> 16) & 0x3FFu) + ((x >> 31) << 15)); }   // |v| in [0.0078, 0.03)
  unsigned short *qkv, *o, *dov, *dqkv;
  float *lse, *delta;
  hipMalloc(&qkv, n * 2); hipMalloc(&dqkv, n * 2); hipMalloc(&o, (size_t)B * T * d * 2); hipMalloc(&dov, (size_t)B * T * d * 2);
  hipMalloc(&lse, (size_t)B * H * T * 4); hipMalloc(&delta, (size_t)B * H * T * 4);
  hipMemcpy(qkv, h.data(), n * 2, hipMemcpyHostToDevice);
  hipMemcpy(dov, h.data(), (size_t)B * T * d * 2, hipMemcpyHostToDevice);
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  const float scale = 0.158113883f;
  auto fwd = [&] { return e4t_attention_fwd(qkv, qkv + d, qkv + 2 * d, o, lse, B, H, T, T, DH, ld, ld, ld, d, (long long)T * ld, (long long)T * ld, (long long)T * ld, (long long)T * d, scale, 0, nullptr); };
  auto bwd = [&] { return e4t_attention_bwd(qkv, qkv + d, qkv + 2 * d, o, dov, lse, delta, dqkv, dqkv + d, dqkv + 2 * d, B, H, T, T, DH, ld, ld, ld, d, (long long)T * ld, (long long)T * ld, (long long)T * ld, (long long)T * d, scale, 0, nullptr); };
  for (int pass = 0; pass < 2; ++pass) {
    auto run = [&] { return pass == 0 ? fwd() : bwd(); };
    for (int i = 0; i < 3; ++i) if (run() != 0) { printf("launch failed\n"); return 1; }
    hipDeviceSynchronize();
    hipEventRecord(e0, nullptr);
    for (int i = 0; i < 10; ++i) run();
    hipEventRecord(e1, nullptr);
    hipEventSynchronize(e1);
    float ms = 0; hipEventElapsedTime(&ms, e0, e1);
    printf("ATTN_PROBE=%d %s %.1f us\n", ATTN_PROBE, pass == 0 ? "fwd" : "bwd(delta+dkv+dq)", ms * 100.f);
  }
  return 0;
}
