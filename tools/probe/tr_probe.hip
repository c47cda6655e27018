// Probe of gfx950's ds_read_b64_tr_b16: which LDS 16-bit element does lane l receive in result slot j, as a function of the
// per-lane byte address it supplied?  LDS holds element index i at element i (u16).  Two address patterns are printed.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>
__global__ void probe(uint16_t* out, int mode) {
  __shared__ uint16_t lds[4096];
  for (int i = threadIdx.x; i < 4096; i += 64) lds[i] = (uint16_t)i;
  __syncthreads();
  const int l = threadIdx.x;
  uint32_t addr;   // byte address inside `lds`
  if (mode == 0) addr = l * 8;                                   // natural: lane l -> elements 4l .. 4l+3
  else if (mode == 1) addr = (l & 15) * 256 + (l >> 4) * 8;      // lane i of a group -> row i (128 elems / row), 4 elems at col 4g
  else addr = (l & 3) * 256 + ((l >> 2) & 3) * 8 + (l >> 4) * 32;  // 4 rows x 4 col-quads per group
  addr += (uint32_t)(uintptr_t)lds;                              // LDS base offset (address space 3 -> low 32 bits)
  uint64_t v;
  asm volatile("ds_read_b64_tr_b16 %0, %1\n s_waitcnt lgkmcnt(0)" : "=v"(v) : "v"(addr) : "memory");
  out[l * 4 + 0] = (uint16_t)(v & 0xffff);
  out[l * 4 + 1] = (uint16_t)((v >> 16) & 0xffff);
  out[l * 4 + 2] = (uint16_t)((v >> 32) & 0xffff);
  out[l * 4 + 3] = (uint16_t)((v >> 48) & 0xffff);
}
int main() {
  uint16_t* d; hipMalloc(&d, 64 * 4 * 2);
  uint16_t h[256];
  for (int mode = 0; mode < 3; ++mode) {
    hipLaunchKernelGGL(probe, dim3(1), dim3(64), 0, 0, d, mode);
    hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost);
    printf("mode %d\n", mode);
    for (int l = 0; l < 64; ++l) printf("  lane %2d: %5d %5d %5d %5d\n", l, h[l * 4], h[l * 4 + 1], h[l * 4 + 2], h[l * 4 + 3]);
  }
  return 0;
}
