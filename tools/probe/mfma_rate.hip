// Calibration for the attention / GEMM analysis: shader clock under MFMA load (clock64 vs the 100 MHz wall_clock64) and the issue
// cost of v_mfma_f32_32x32x16_bf16 in a dependent accumulate chain vs NACC independent accumulators, at 1 / 2 / 4 waves per SIMD,
// alone and with VALU work (v_exp_f32 / v_fma_f32) interleaved in the same wave or running in OTHER waves of the SIMD.
//   hipcc --offload-arch=gfx950 -O3 tools/probe/mfma_rate.hip -o tools/probe/bin/mfma_rate
#include <hip/hip_runtime.h>
#include <cstdio>
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(16))) float f32x16;

// MODE 0: MFMA only.  1: MFMA wave-streams, plus NV VALU fmas per MFMA in the same wave.  2: even waves MFMA-only, odd waves VALU-only.
template <int NACC, int MODE, int NV, bool EXP>
__global__ __launch_bounds__(512) void k(float* out, long long* stats, int iters) {
  f32x16 acc[NACC];
  for (int a = 0; a < NACC; ++a) for (int r = 0; r < 16; ++r) acc[a][r] = 0.f;
  bf16x8 x, y;
  for (int j = 0; j < 8; ++j) { x[j] = (__bf16)(threadIdx.x * 0.001f); y[j] = (__bf16)(j * 0.01f); }
  float v[8];
  for (int j = 0; j < 8; ++j) v[j] = threadIdx.x * 0.01f + j;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int role = (wave >> 2) & 1;      // waves w and w + 4 share a SIMD
  const bool do_mfma = MODE != 2 || role == 0, do_valu = MODE == 1 || (MODE == 2 && role == 1);
  __syncthreads();
  const long long c0 = clock64(), w0 = wall_clock64();
  if (MODE >= 2) {                     // role 0: a pure MFMA stream; role 1: a pure VALU stream of the same instruction count x NV
    if (role == 0) {
      if (MODE != 3)
      for (int i = 0; i < iters; ++i)
#pragma unroll
        for (int u = 0; u < 8; ++u) acc[u % NACC] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(x, y, acc[u % NACC], 0, 0, 0);
    } else {
      if (MODE != 4)
      for (int i = 0; i < iters; ++i)
#pragma unroll
        for (int u = 0; u < 8 * NV; ++u) {
          if (EXP) asm volatile("v_exp_f32 %0, %0" : "+v"(v[u % 8]));
          else asm volatile("v_fma_f32 %0, %0, %1, %1" : "+v"(v[u % 8]) : "v"(1.0001f));
        }
    }
  } else
  for (int i = 0; i < iters; ++i) {
#pragma unroll
    for (int u = 0; u < 8; ++u) {
      if (do_mfma) acc[u % NACC] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(x, y, acc[u % NACC], 0, 0, 0);
      if (do_valu) {
#pragma unroll
        for (int n = 0; n < NV; ++n) {
          if (EXP) asm volatile("v_exp_f32 %0, %0" : "+v"(v[n % 8]));
          else asm volatile("v_fma_f32 %0, %0, %1, %1" : "+v"(v[n % 8]) : "v"(1.0001f));
        }
      }
    }
  }
  const long long c1 = clock64(), w1 = wall_clock64();
  float s = 0.f;
  for (int a = 0; a < NACC; ++a) for (int r = 0; r < 16; ++r) s += acc[a][r];
  for (int j = 0; j < 8; ++j) s += v[j];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
  if (threadIdx.x == 0 && blockIdx.x == 0) { stats[0] = c1 - c0; stats[1] = w1 - w0; }
}

template <int NACC, int MODE, int NV, bool EXP>
void run(const char* name, int threads, int blocks_per_cu, float* out, long long* stats) {
  const int iters = 4000;
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  hipLaunchKernelGGL((k<NACC, MODE, NV, EXP>), dim3(256 * blocks_per_cu), dim3(threads), 0, 0, out, stats, iters);   // warm
  hipEventRecord(e0, 0);
  hipLaunchKernelGGL((k<NACC, MODE, NV, EXP>), dim3(256 * blocks_per_cu), dim3(threads), 0, 0, out, stats, iters);
  hipEventRecord(e1, 0);
  hipEventSynchronize(e1);
  float ms = 0; hipEventElapsedTime(&ms, e0, e1);
  long long h[2];
  hipMemcpy(h, stats, 16, hipMemcpyDeviceToHost);
  const double mhz = (double)h[0] / (double)h[1] * 100.0;
  const int wps = threads / 256 * blocks_per_cu;                       // waves per SIMD
  const int mfma_wps = MODE >= 2 ? wps / 2 : wps;                       // of which issue MFMAs
  const double clk_total = ms * 1e-3 * mhz * 1e6;                       // kernel duration in shader clocks
  printf("%-52s %d waves/SIMD: kernel %8.1f us = %6.1f clk per MFMA issued on a SIMD (wave 0 saw %5.1f per own MFMA), %4.0f MHz\n", name, wps,
         ms * 1e3, clk_total / ((double)iters * 8 * mfma_wps), (double)h[0] / iters / 8, mhz);
}

int main() {
  float* out; long long* stats;
  hipMalloc(&out, 256 * 8 * 1024 * 4); hipMalloc(&stats, 16);
  // one workgroup of 256 threads per CU = 1 wave per SIMD; 512 threads = 2 per SIMD; 2 blocks x 512 = 4 per SIMD
  run<1, 0, 0, false>("MFMA dependent chain (1 accumulator)", 256, 1, out, stats);
  run<2, 0, 0, false>("MFMA 2 accumulators", 256, 1, out, stats);
  run<4, 0, 0, false>("MFMA 4 accumulators", 256, 1, out, stats);
  run<1, 0, 0, false>("MFMA dependent chain (1 accumulator)", 512, 1, out, stats);
  run<1, 0, 0, false>("MFMA dependent chain (1 accumulator)", 512, 2, out, stats);
  run<4, 0, 0, false>("MFMA 4 accumulators", 512, 2, out, stats);
  run<4, 1, 4, false>("MFMA 4 acc + 4 v_fma per MFMA, same wave", 256, 1, out, stats);
  run<4, 1, 8, false>("MFMA 4 acc + 8 v_fma per MFMA, same wave", 256, 1, out, stats);
  run<4, 1, 16, false>("MFMA 4 acc + 16 v_fma per MFMA, same wave", 256, 1, out, stats);
  run<4, 1, 4, true>("MFMA 4 acc + 4 v_exp per MFMA, same wave", 256, 1, out, stats);
  run<4, 1, 8, true>("MFMA 4 acc + 8 v_exp per MFMA, same wave", 256, 1, out, stats);
  // waves w (MFMA stream) and w + 4 (VALU stream) of a 512-thread workgroup share a SIMD
  run<4, 4, 8, false>("SIMD pair: MFMA wave alone (partner idle)", 512, 1, out, stats);
  run<4, 3, 8, false>("SIMD pair: VALU wave alone, 8 v_fma per MFMA slot", 512, 1, out, stats);
  run<4, 2, 8, false>("SIMD pair: MFMA wave | 8 v_fma per slot wave", 512, 1, out, stats);
  run<4, 3, 16, false>("SIMD pair: VALU wave alone, 16 v_fma per slot", 512, 1, out, stats);
  run<4, 2, 16, false>("SIMD pair: MFMA wave | 16 v_fma per slot wave", 512, 1, out, stats);
  run<4, 3, 4, true>("SIMD pair: VALU wave alone, 4 v_exp per slot", 512, 1, out, stats);
  run<4, 2, 4, true>("SIMD pair: MFMA wave | 4 v_exp per slot wave", 512, 1, out, stats);
  run<4, 2, 8, false>("2 SIMD pairs (4 waves/SIMD): MFMA | 8 v_fma", 512, 2, out, stats);
  return 0;
}
