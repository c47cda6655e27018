// Probe: how fast can a CU stage L2-resident operand panels into LDS on gfx950?
// Every GEMM kernel of this library (128x128 / 128x160 two-stage, 256x256 ping-pong, the persistent 256xBN experiment, the
// guide's 8-phase template) ends up moving ~18-26 B/clk/CU through the HBM/L2 -> LDS path whatever its schedule: this measures the
// ceiling of that path in isolation, for the access shapes the kernels use, so that tile shapes are chosen against a known number.
//   build: hipcc --offload-arch=gfx950 -O3 -o dma_bw dma_bw.hip        run: ./dma_bw
// Each workgroup "owns" a panel of ROWS rows x K bf16 (row stride ld) and streams it K-tile by K-tile into LDS, over and over
// (the panel set of all workgroups is L2-resident); no MFMA, no ds_read unless asked.  Variants:
//   mode 0: LDS-DMA (buffer_load ... lds), pieces of 8 rows x 128 B   (64-wide K-tiles)
//   mode 1: LDS-DMA, pieces of 16 rows x 64 B                        (32-wide K-tiles)
//   mode 2: global_load_dwordx4 -> VGPR (no LDS), 8 rows x 128 B per wave-instruction
//   mode 3: mode 2 + ds_write_b128 of the loaded data
//   mode 4: mode 0 + a ds_read_b128 stream of the same volume x RD from the LDS (what the MFMA fragment reads cost)
//   mode 5: mode 4 without the workgroup barrier (every wave reads only what it staged itself)    mode 6: mode 0 + the barrier, no reads
//   mode 7: mode 4 + RD*PIECES*2/3 MFMAs per tile (the arithmetic of a 128x128x64 tile per wave) fed from the values read
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <vector>

typedef uint16_t bf16_t;

__device__ __forceinline__ void buf_dma16(__amdgpu_buffer_rsrc_t rs, unsigned voff, int soff, bf16_t* lds) {
  __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, (__attribute__((address_space(3))) void*)lds, 16, voff, soff, 0, 0);
}
template <int N>
__device__ __forceinline__ void wait_vmcnt() { asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory"); }

// NW waves per workgroup; PIECES 1-KiB pieces per wave per K-tile; DEPTH K-tiles in flight; the LDS ring holds DEPTH + 1 tiles
template <int MODE, int NW, int PIECES, int DEPTH, int RD>
__global__ __launch_bounds__(NW * 64) void stream_kernel(const bf16_t* A, int ld, int K, int rows_per_wg, int iters, unsigned bytes, float* sink, int npanels) {
  constexpr int KT = (MODE == 1) ? 32 : 64;
  constexpr int RPP = 512 / KT;                          // rows per piece
  constexpr int TILE = NW * PIECES * 512;                // elements per K-tile in LDS
  extern __shared__ __attribute__((aligned(16))) bf16_t smem[];
  const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int lrow = lane / (KT / 8), lslot = lane % (KT / 8);
  const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc((void*)A, 0, (int)bytes, 0x00020000);
  unsigned vo[PIECES];
#pragma unroll
  for (int i = 0; i < PIECES; ++i) {
    const int r = (npanels ? (int)(blockIdx.x % npanels) : (int)blockIdx.x) * rows_per_wg + (wave * PIECES + i) * RPP + lrow;
    const int sw = KT == 64 ? ((r >> 1) & 7) : ((r >> 2) & 3);
    vo[i] = (unsigned)(((size_t)r * ld + ((lslot ^ sw) * 8)) * 2);
  }
  const int nkt = K / KT;
  float acc = 0.f;
  uint4 regs[PIECES];
  int slot = 0;
  auto issue = [&](int kt, int s) {
#pragma unroll
    for (int i = 0; i < PIECES; ++i) {
      if (MODE == 0 || MODE == 1 || MODE >= 4) buf_dma16(rs, vo[i], kt * KT * 2, smem + s * TILE + (wave * PIECES + i) * 512);
    }
  };
  const int total = iters * nkt;
  if (MODE == 2 || MODE == 3) {
    for (int t = 0; t < total; ++t) {
      const int kt = t % nkt;
#pragma unroll
      for (int i = 0; i < PIECES; ++i) {
        const unsigned off = vo[i] + kt * KT * 2;
        regs[i] = *(const uint4*)((const char*)A + off);
      }
      if (MODE == 3) {
#pragma unroll
        for (int i = 0; i < PIECES; ++i) *(uint4*)(smem + slot * TILE + (wave * PIECES + i) * 512 + lane * 8) = regs[i];
        slot = slot + 1 > DEPTH ? 0 : slot + 1;
      } else {
#pragma unroll
        for (int i = 0; i < PIECES; ++i) acc += __uint_as_float(regs[i].x ^ regs[i].w);
      }
    }
  } else {
    int is = 0;
    for (int t = 0; t < DEPTH && t < total; ++t) { issue(t % nkt, is); is = is + 1 > DEPTH ? 0 : is + 1; }
    for (int t = 0; t < total; ++t) {
      if (t + DEPTH < total) {
        issue((t + DEPTH) % nkt, is);
        is = is + 1 > DEPTH ? 0 : is + 1;
        wait_vmcnt<PIECES * DEPTH>();
      } else {
        wait_vmcnt<0>();
      }
      if (MODE == 4 || MODE == 6 || MODE == 7) __builtin_amdgcn_s_barrier();
      if (MODE == 4 || MODE == 5 || MODE == 7) {
        uint4 v[RD * PIECES > 0 ? RD * PIECES : 1];
#pragma unroll
        for (int r = 0; r < RD * PIECES; ++r)
          v[r] = *(const uint4*)(smem + slot * TILE + ((wave * PIECES + (r % PIECES)) * 512 + ((lane * 8 + r * 64) & 511)));
        if (MODE == 7) {
          typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
          typedef __attribute__((ext_vector_type(16))) float f32x16;
          static_assert(sizeof(bf16x8) == sizeof(uint4), "");
          f32x16 c0 = {}, c1 = {};
#pragma unroll
          for (int r = 0; r < RD * PIECES; r += 3) {      // 12 reads -> 8 MFMAs (2 per 3 reads)
            bf16x8 a, b0, b1;
            __builtin_memcpy(&a, &v[r], 16); __builtin_memcpy(&b0, &v[(r + 1) % (RD * PIECES)], 16); __builtin_memcpy(&b1, &v[(r + 2) % (RD * PIECES)], 16);
            c0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b0, c0, 0, 0, 0);
            c1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b1, c1, 0, 0, 0);
          }
          acc += c0[0] + c1[3];
        } else {
#pragma unroll
          for (int r = 0; r < RD * PIECES; ++r) acc += __uint_as_float(v[r].x ^ v[r].w);
        }
        slot = slot + 1 > DEPTH ? 0 : slot + 1;
      }
    }
  }
  if (acc == 123.456f) sink[0] = acc;
}

template <int MODE, int NW, int PIECES, int DEPTH, int RD = 0>
void run(const char* name, const bf16_t* A, int ld, int K, int wgs_per_cu, float* sink, unsigned bytes, int ncu, int npanels = 64) {
  const int rows_per_wg = NW * PIECES * ((MODE == 1) ? 16 : 8);
  const int grid = ncu * wgs_per_cu;
  const int iters = 200;
  const size_t lds = (size_t)(DEPTH + 1) * NW * PIECES * 1024;
  if ((size_t)(npanels ? npanels : grid) * rows_per_wg * ld * 2 > bytes) { printf("%-58s skipped (panel set larger than the buffer)\n", name); return; }
  hipFuncSetAttribute((const void*)stream_kernel<MODE, NW, PIECES, DEPTH, RD>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
  hipEvent_t e0, e1;
  hipEventCreate(&e0); hipEventCreate(&e1);
  for (int rep = 0; rep < 2; ++rep) {
    hipEventRecord(e0);
    hipLaunchKernelGGL((stream_kernel<MODE, NW, PIECES, DEPTH, RD>), dim3(grid), dim3(NW * 64), lds, 0, A, ld, K, rows_per_wg, iters, bytes, sink, npanels);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
  }
  float ms = 0;
  hipEventElapsedTime(&ms, e0, e1);
  const double total = (double)grid * rows_per_wg * K * 2.0 * iters;
  const double tbs = total / (ms * 1e-3) / 1e12;
  printf("%-58s ld %5d K %5d  %d WG/CU x %2d waves, %d KiB in flight/CU : %7.2f TB/s = %5.1f B/clk/CU @2.1GHz  (%.3f ms, L2 set %.1f MB)\n", name, ld, K, wgs_per_cu, NW,
         (int)(wgs_per_cu * NW * PIECES * DEPTH), tbs, tbs * 1e12 / ncu / 2.1e9, ms, (double)(npanels ? npanels : grid) * rows_per_wg * ld * 2 / 1e6);
  hipEventDestroy(e0); hipEventDestroy(e1);
}

int main() {
  int ncu = 256;
  hipDeviceGetAttribute(&ncu, hipDeviceAttributeMultiprocessorCount, 0);
  const size_t bytes = 512u << 20;
  bf16_t* A; float* sink;
  hipMalloc(&A, bytes); hipMalloc(&sink, 64);
  hipMemset(A, 0x3c, bytes);
  printf("CUs: %d\n", ncu);
  for (int K : {320, 1280}) {
    const int ld = K;
    // LDS-DMA, 64-wide K-tiles
    run<0, 8, 2, 1>("dma 8x128B  8 waves x 2 pieces, 1 tile ahead", A, ld, K, 1, sink, (unsigned)bytes, ncu);
    run<0, 8, 2, 1>("dma 8x128B  8 waves x 2 pieces, 1 tile ahead", A, ld, K, 2, sink, (unsigned)bytes, ncu);
    run<0, 8, 2, 2>("dma 8x128B  8 waves x 2 pieces, 2 tiles ahead", A, ld, K, 2, sink, (unsigned)bytes, ncu);
    run<0, 8, 4, 2>("dma 8x128B  8 waves x 4 pieces, 2 tiles ahead", A, ld, K, 1, sink, (unsigned)bytes, ncu);
    run<0, 8, 4, 3>("dma 8x128B  8 waves x 4 pieces, 3 tiles ahead", A, ld, K, 1, sink, (unsigned)bytes, ncu);
    run<0, 4, 4, 2>("dma 8x128B  4 waves x 4 pieces, 2 tiles ahead", A, ld, K, 2, sink, (unsigned)bytes, ncu);
    run<0, 4, 8, 2>("dma 8x128B  4 waves x 8 pieces, 2 tiles ahead", A, ld, K, 1, sink, (unsigned)bytes, ncu);
    // LDS-DMA, 32-wide K-tiles
    run<1, 8, 2, 2>("dma 16x64B  8 waves x 2 pieces, 2 tiles ahead", A, ld, K, 2, sink, (unsigned)bytes, ncu);
    run<1, 8, 3, 4>("dma 16x64B  8 waves x 3 pieces, 4 tiles ahead", A, ld, K, 1, sink, (unsigned)bytes, ncu);
    run<1, 8, 4, 3>("dma 16x64B  8 waves x 4 pieces, 3 tiles ahead", A, ld, K, 1, sink, (unsigned)bytes, ncu);
    // plain loads
    run<2, 8, 8, 1>("global_load_dwordx4 -> VGPR 8x128B, 8 waves x 8", A, ld, K, 2, sink, (unsigned)bytes, ncu);
    run<3, 8, 4, 1>("global_load_dwordx4 -> VGPR -> ds_write_b128, 8 x 4", A, ld, K, 2, sink, (unsigned)bytes, ncu);
    // the same with one distinct panel per workgroup (set larger than the L2s: served by the Infinity Cache / HBM)
    run<0, 8, 4, 2>("dma 8x128B  8 waves x 4 pieces, 2 ahead, DISTINCT panels", A, ld, K, 1, sink, (unsigned)bytes, ncu, 0);
    run<1, 8, 3, 4>("dma 16x64B  8 waves x 3 pieces, 4 ahead, DISTINCT panels", A, ld, K, 1, sink, (unsigned)bytes, ncu, 0);
    run<2, 8, 8, 1>("global_load_dwordx4 -> VGPR 8 waves x 8, DISTINCT panels", A, ld, K, 2, sink, (unsigned)bytes, ncu, 0);
    // LDS-DMA with concurrent fragment-read traffic: what costs — the reads, the barrier, the MFMAs?
    run<6, 8, 2, 2, 0>("dma 8x128B 8x2, 2 ahead + barrier only", A, ld, K, 2, sink, (unsigned)bytes, ncu);
    run<5, 8, 2, 2, 3>("dma 8x128B 8x2, 2 ahead + reads x3, NO barrier", A, ld, K, 2, sink, (unsigned)bytes, ncu);
    run<4, 8, 2, 2, 1>("dma 8x128B 8x2, 2 ahead + barrier + reads x1", A, ld, K, 2, sink, (unsigned)bytes, ncu);
    run<4, 8, 2, 2, 3>("dma 8x128B 8x2, 2 ahead + barrier + reads x3", A, ld, K, 2, sink, (unsigned)bytes, ncu);
    run<4, 8, 2, 2, 6>("dma 8x128B 8x2, 2 ahead + barrier + reads x6", A, ld, K, 2, sink, (unsigned)bytes, ncu);
    run<4, 8, 2, 1, 3>("dma 8x128B 8x2, 1 ahead + barrier + reads x3", A, ld, K, 2, sink, (unsigned)bytes, ncu);
    run<4, 8, 4, 2, 3>("dma 8x128B 8x4, 2 ahead + barrier + reads x3", A, ld, K, 1, sink, (unsigned)bytes, ncu);
    run<7, 8, 2, 1, 6>("dma 8x128B 8x2, 1 ahead + barrier + reads x6 + 8 MFMA (~128^2 tile)", A, ld, K, 2, sink, (unsigned)bytes, ncu);
    run<7, 8, 2, 2, 6>("dma 8x128B 8x2, 2 ahead + barrier + reads x6 + 8 MFMA", A, ld, K, 2, sink, (unsigned)bytes, ncu);
    run<7, 8, 4, 2, 3>("dma 8x128B 8x4, 2 ahead + barrier + reads x3 + 8 MFMA", A, ld, K, 1, sink, (unsigned)bytes, ncu);
  }
  return 0;
}
