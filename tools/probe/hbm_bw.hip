// What does a streaming kernel reach on this chip, and with which launch shape?  Copy (1 read + 1 write stream), read-only (sum) and
// 3-in-1-out (the GEGLU backward / AdamW pattern) over buffers far larger than L2 + MALL, for: bytes in flight per thread (U x 16 B loads
// issued before the first use), workgroups per CU (grid = 256 x G, grid-stride), block size, non-temporal loads / stores.
//   hipcc --offload-arch=gfx950 -O3 tools/probe/hbm_bw.hip -o tools/probe/bin/hbm_bw && tools/probe/bin/hbm_bw
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef __attribute__((ext_vector_type(4))) unsigned u32x4;

template <int U, bool NT>
__global__ __launch_bounds__(256) void copy_k(const u32x4* __restrict__ in, u32x4* __restrict__ out, size_t n) {
  const size_t stride = (size_t)gridDim.x * blockDim.x;
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride * U) {
    u32x4 v[U];
#pragma unroll
    for (int u = 0; u < U; ++u) { const size_t j = i + u * stride; if (j < n) v[u] = NT ? __builtin_nontemporal_load(in + j) : in[j]; }
#pragma unroll
    for (int u = 0; u < U; ++u) { const size_t j = i + u * stride; if (j < n) { if (NT) __builtin_nontemporal_store(v[u], out + j); else out[j] = v[u]; } }
  }
}
template <int U>
__global__ __launch_bounds__(256) void read_k(const u32x4* __restrict__ in, unsigned* out, size_t n) {
  const size_t stride = (size_t)gridDim.x * blockDim.x;
  unsigned s = 0;
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride * U) {
    u32x4 v[U];
#pragma unroll
    for (int u = 0; u < U; ++u) { const size_t j = i + u * stride; v[u] = j < n ? in[j] : u32x4{0, 0, 0, 0}; }
#pragma unroll
    for (int u = 0; u < U; ++u) s += v[u].x ^ v[u].y ^ v[u].z ^ v[u].w;
  }
  if (s == 0x12345) out[0] = s;
}
template <int U>
__global__ __launch_bounds__(256) void write_k(u32x4* out, size_t n) {
  const size_t stride = (size_t)gridDim.x * blockDim.x;
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) out[i] = u32x4{(unsigned)i, 1, 2, 3};
}
// p, m, v read + written, g read: AdamW's 7 streams (28 B per fp32 element) with a little arithmetic
template <int U>
__global__ __launch_bounds__(256) void adam_k(u32x4* p, const u32x4* g, u32x4* m, u32x4* v, size_t n) {
  const size_t stride = (size_t)gridDim.x * blockDim.x;
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride * U) {
    u32x4 P[U], G[U], M[U], V[U];
#pragma unroll
    for (int u = 0; u < U; ++u) { const size_t j = i + u * stride < n ? i + u * stride : n - 1; P[u] = p[j]; G[u] = g[j]; M[u] = m[j]; V[u] = v[j]; }
#pragma unroll
    for (int u = 0; u < U; ++u) { const size_t j = i + u * stride; if (j < n) { M[u] = M[u] + G[u]; V[u] = V[u] ^ G[u]; P[u] = P[u] + M[u]; p[j] = P[u]; m[j] = M[u]; v[j] = V[u]; } }
  }
}

template <class F>
float time_ms(F f, int iters = 5) {
  hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
  f(); hipDeviceSynchronize();
  hipEventRecord(a);
  for (int i = 0; i < iters; ++i) f();
  hipEventRecord(b); hipEventSynchronize(b);
  float ms; hipEventElapsedTime(&ms, a, b);
  return ms / iters;
}

int main() {
  const size_t bytes = (size_t)1536 << 20;          // 1.5 GB per buffer
  const size_t n = bytes / 16;
  u32x4 *a, *b, *c, *d; unsigned* o;
  hipMalloc(&a, bytes); hipMalloc(&b, bytes); hipMalloc(&c, bytes); hipMalloc(&d, bytes); hipMalloc(&o, 64);
  hipMemset(a, 1, bytes); hipMemset(b, 2, bytes); hipMemset(c, 3, bytes); hipMemset(d, 4, bytes);
  printf("# 1.5 GB buffers; GB/s = bytes moved / time\n");
  for (int G : {4, 8, 16, 32, 64}) {
    const int grid = 256 * G;
#define COPY(U, NT) printf("copy  U=%d nt=%d grid=256x%-2d  %7.0f GB/s\n", U, NT, G, 2.0 * bytes / 1e6 / time_ms([&] { hipLaunchKernelGGL((copy_k<U, NT>), dim3(grid), dim3(256), 0, 0, a, b, n); }));
    COPY(1, false) COPY(2, false) COPY(4, false) COPY(8, false) COPY(4, true)
#define READ(U) printf("read  U=%d      grid=256x%-2d  %7.0f GB/s\n", U, G, 1.0 * bytes / 1e6 / time_ms([&] { hipLaunchKernelGGL((read_k<U>), dim3(grid), dim3(256), 0, 0, a, o, n); }));
    READ(1) READ(4) READ(8)
    printf("write         grid=256x%-2d  %7.0f GB/s\n", G, 1.0 * bytes / 1e6 / time_ms([&] { hipLaunchKernelGGL((write_k<1>), dim3(grid), dim3(256), 0, 0, b, n); }));
#define ADAM(U) printf("adam7 U=%d      grid=256x%-2d  %7.0f GB/s\n", U, G, 7.0 * bytes / 1e6 / time_ms([&] { hipLaunchKernelGGL((adam_k<U>), dim3(grid), dim3(256), 0, 0, a, b, c, d, n); }));
    ADAM(1) ADAM(2) ADAM(4)
  }
  return 0;
}
