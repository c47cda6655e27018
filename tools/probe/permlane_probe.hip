#include <hip/hip_runtime.h>
#include <cstdio>
__device__ __forceinline__ float xhalf_max(float v) {
  float a = v, b = v;
  asm volatile("s_nop 1\n\tv_permlane32_swap_b32 %0, %1" : "+v"(a), "+v"(b));
  return fmaxf(a, b);
}
__global__ void k(float* x) {
  x[threadIdx.x] = xhalf_max(x[threadIdx.x]);
}
int main() {
  float h[64], *d; for (int i = 0; i < 64; ++i) h[i] = (float)((i * 37) % 64);
  hipMalloc(&d, 256); hipMemcpy(d, h, 256, hipMemcpyHostToDevice);
  hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, d);
  float o[64]; hipMemcpy(o, d, 256, hipMemcpyDeviceToHost);
  int bad = 0; for (int i = 0; i < 64; ++i) { float w = fmaxf(h[i % 32], h[32 + i % 32]); if (o[i] != w) ++bad; }
  printf("permlane32_swap cross-half max: %d mismatches\n", bad);
  return bad != 0;
}
