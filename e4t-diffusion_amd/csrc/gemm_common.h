// Pieces shared by the GEMM translation units (gemm.hip, gemm_ps.hip): argument block, fused epilogue, LDS-DMA helpers,
// XCD-aware tile raster.  Everything lives in an anonymous namespace: each translation unit gets its own copy.
#pragma once
#include "common.h"
#include "../../include/e4t_hip.h"
#include <stdlib.h>
#include <type_traits>

struct GemmArgs {
  // A operand
  const bf16_t* A;
  const bf16_t* A2;
  int K1;  // columns [0,K1) come from A, [K1,K) from A2 (K1 == K when A2 == nullptr)
  int lda, lda2;
  // conv geometry (MODE != 0)
  int Hin, Win, Cin, Hout, Wout, mode;
  // B operand [N][K]
  const bf16_t* B;
  int ldb;
  // output / epilogue
  void* C;
  int ldc;
  const float* bias;
  const void* residual;
  int ldr;
  const float* rowbias;
  int rows_per_batch, ldrb;
  int M, N, K;
  float alpha;
  int flags;
  // split-K / batch
  float* ws;
  int ktiles_per_split;
  int splitk;       // splits per batch entry (grid.z = batch * splitk)
  int group_m;      // row panels per raster group (xcd_tile)
  int reduce_batch; // partials of all (batch, split) pairs are summed into ONE C
  int fast_epi;     // bf16 C, 16-byte aligned rows: LDS-staged vectorised epilogue
  int fast_f32;     // fp32 C (+ fp32 residual), no accumulate, no row bias: direct line-wide stores from the accumulator layout
  int ps_pre;       // persistent kernel: bias / row bias prefetched into LDS by DMA (write_tile<..., PRE>)
  int chan_major;   // stride-1 3x3 conv: K walked channel-chunk-major (cm_step) instead of tap-major
  int xcd3;         // TN kernel with split-K: workgroups re-dealt over (split, tile) so one XCD owns whole K-slices (xcd_tile3)
  // Row panels (gemm_pq_kernel only): logical row r of A / C / residual lives at physical row (r / panel_rows) * panel_stride + panel_off +
  // r % panel_rows.  panel_rows = 0: dense.  Lets the 16 x 256 patch tokens of a [16][257] ViT token matrix run as 16 full 256-row tiles.
  int panel_rows, panel_stride, panel_off;
  int zslab;        // partial-slab index of this workgroup when it is not blockIdx.z (set in-kernel by a re-dealing kernel; host: -1)
  // Tail rows (dense GEMMs, round 5): the tile grid covers rows [0, M) and the few rows [tail_row0, tail_row0 + tail_rows) behind it
  // (tail_rows <= 32, 0 = none) are computed by gemm_tail() at the end of the same launch — see there.
  int tail_row0, tail_rows;
  long long strideA, strideB, strideC, strideBias;
  float* colstats;                       // optional [M/32][N][2] column statistics of the output (fast bf16 epilogue only)
  unsigned a_bytes, a2_bytes, b_bytes;   // operand extents from the (batch-adjusted) base pointers, for buffer resources (pp kernel)
};

namespace {

constexpr int BK = 64;        // K-tile (bf16 elements)
constexpr int LDS_LD = BK + 8;  // padded LDS row stride (elements): 144 B


template <bool GELU_OK = true>
__device__ __forceinline__ void epilogue_store(const GemmArgs& p, float v, int row, int col) {
  v *= p.alpha;
  if (p.bias) v += p.bias[col];
  if (p.rowbias) v += p.rowbias[(size_t)(row / p.rows_per_batch) * p.ldrb + col];
  if constexpr (GELU_OK) { if (p.flags & E4T_ACT_GELU) v = gelu_f(v); }
  if (p.residual) {
    if (p.flags & E4T_RES_F32) v += ((const float*)p.residual)[(size_t)row * p.ldr + col];
    else v += bf2f(((const bf16_t*)p.residual)[(size_t)row * p.ldr + col]);
  }
  if (p.flags & E4T_OUT_F32) {
    float* c = (float*)p.C + (size_t)row * p.ldc + col;
    if (p.flags & E4T_ACCUM) v += *c;
    *c = v;
  } else {
    bf16_t* c = (bf16_t*)p.C + (size_t)row * p.ldc + col;
    if (p.flags & E4T_ACCUM) v += bf2f(*c);
    *c = f2bf(v);
  }
}

// LDS staging area of wave `wave` for write_tile<WM, WN, ...> when the waves' areas are packed back to back from `smem`.
template <int WM, int WN>
__device__ __forceinline__ bf16_t* wave_stage(bf16_t* smem, int wave) { return smem + wave * (WM * (WN + 8)); }

// Write one wave's WM x WN accumulator tile (origin mw, nw) with the fused epilogue.
#ifdef DMA_TRACE
#define WT_STAMP(k) do { if (dt_ptr) dt_ptr[k] = __builtin_readcyclecounter(); } while (0)
#else
#define WT_STAMP(k) do { } while (0)
#endif
// GENERAL: the epilogue variant with the exact-GELU activation and the per-row row-bias lookup (rows_per_batch not a multiple of
// 32).  It is a separate INSTANTIATION, not a branch: inlined next to the plain path its erff expansion over 16 x FM x FN
// elements set the register allocation of the whole kernel (288 instead of 208 registers in the 128 x 160 tile = one
// workgroup per CU instead of two).  The launcher picks the variant (launch_gemm).
// PRE (persistent kernel, gemm_ps.hip): bias and row bias of the tile's columns were brought into LDS ahead of time (pre_bias /
// pre_rb point at the entry of column nw; one row-bias row per tile) and the staging barrier is a raw s_barrier: no global load
// and no compiler-placed vmcnt(0) sits between the K loop and the staging writes, so the operand DMA of the NEXT tile, in flight
// at this point, is not drained in front of the epilogue (vmcnt is an in-order counter: waiting for a load issued here means
// waiting for every DMA issued before it).
template <int WM, int WN, int FM, int FN, bool GENERAL = false, bool PRE = false>
__device__ __forceinline__ void write_tile(const GemmArgs& p, f32x16 (&acc)[FM][FN], bf16_t* stage, int lane, int mw, int nw,
                                           unsigned long long* dt_ptr = nullptr, const float* pre_bias = nullptr, const float* pre_rb = nullptr) {
  const int frow = lane & 31, fhi = lane >> 5;
  // ---- epilogue: C/D layout of 32x32 MFMA: col = lane&31, row = (r&3) + 8*(r>>2) + 4*(lane>>5) ----
  const bool partial = p.ws != nullptr;
  if (!partial && p.fast_epi) {
    // bf16 output: stage the wave's WM x WN tile through LDS (the operand tiles are dead after the loop's final
    // barrier) so that C is written — and the residual read — as 16-byte chunks, 128 B contiguous per row.
    // alpha, bias, row bias and GELU are applied in fp32 before the bf16 rounding; the residual is added to the
    // rounded value in fp32 and rounded again, which is exactly what a bf16 linear followed by a bf16 add does.
    constexpr int ELD = WN + 8;          // `stage`: this wave's own WM x ELD staging area in LDS (see wave_stage())
    // Everything the epilogue reads from global memory is fetched by UNCONDITIONAL loads issued back to back (indices clamped
    // into range, values masked afterwards).  The first version guarded each load (`col < N ? bias[col] : 0`, the row-bias
    // inside the 16-element loop, the residual chunk inside the store loop): the compiler answered every guarded load with its
    // own s_waitcnt vmcnt(0) — FN serialized L2 round trips for the bias, 16 x FM x FN for the ResBlock time-embedding row-bias
    // (85 in the 128 x 160 conv tile), one per 16-byte residual chunk (10) — in the epilogue of EVERY workgroup (ISA, round 2).
    float bv[FN], rbv[FM][FN];
    const bool rb_blocked = p.rowbias && (p.rows_per_batch % 32 == 0);        // a 32-row fragment lies inside one batch entry
    if constexpr (PRE) {
#pragma unroll
      for (int j = 0; j < FN; ++j) {
        bv[j] = pre_bias[j * 32 + frow];
        const float rb = pre_rb[j * 32 + frow];
#pragma unroll
        for (int i = 0; i < FM; ++i) rbv[i][j] = rb;
      }
    } else {
#pragma unroll
      for (int j = 0; j < FN; ++j) {
        const int col = min(nw + j * 32 + frow, p.N - 1);
        bv[j] = p.bias ? p.bias[col] : 0.f;
#pragma unroll
        for (int i = 0; i < FM; ++i) {
          const int row = min(mw + i * 32, p.M - 1);
          rbv[i][j] = rb_blocked ? p.rowbias[(size_t)(row / p.rows_per_batch) * p.ldrb + col] : 0.f;
        }
      }
    }
    WT_STAMP(13);
    // The uniform special cases (GELU epilogue, per-row row-bias lookup) are decided ONCE, outside the 16 x FM x FN element loop: as
    // per-element `if`s they were two scalar branches per element — the staging of a 32 x 160 wave tile took 8300 of the
    // workgroup's 32000 cycles on the K = 320 projections (cycle stamps, tools/dma_trace.sh), 1400 without them.
    const bool rb_slow = !PRE && p.rowbias && !rb_blocked;
    if constexpr (!GENERAL) {
#pragma unroll
      for (int i = 0; i < FM; ++i)
#pragma unroll
        for (int j = 0; j < FN; ++j) {
          const int cl = j * 32 + frow;
          const float add = bv[j] + rbv[i][j];
#pragma unroll
          for (int r = 0; r < 16; ++r) {
            const int rl = i * 32 + (r & 3) + 8 * (r >> 2) + 4 * fhi;
            stage[rl * ELD + cl] = f2bf(acc[i][j][r] * p.alpha + add);
          }
          // one fragment at a time: without the fence the scheduler pulls the accumulator reads of ALL fragments (80 AGPR -> VGPR
          // copies in the 128 x 160 tile) in front of the first write, and the kernel loses one of its two waves per SIMD
          __builtin_amdgcn_sched_barrier(0);
        }
    } else {
      const bool gelu = (p.flags & E4T_ACT_GELU) != 0;
#pragma unroll
      for (int i = 0; i < FM; ++i)
#pragma unroll
        for (int j = 0; j < FN; ++j) {
          const int cl = j * 32 + frow;
          const int col = nw + cl;
          const float add = bv[j] + rbv[i][j];
#pragma unroll
          for (int r = 0; r < 16; ++r) {
            const int rl = i * 32 + (r & 3) + 8 * (r >> 2) + 4 * fhi;
            float v = acc[i][j][r] * p.alpha + add;
            if (rb_slow) {      // odd geometry (rows_per_batch not a multiple of 32): per-row lookup
              const int row = mw + rl;
              if (row < p.M && col < p.N) v += p.rowbias[(size_t)(row / p.rows_per_batch) * p.ldrb + col];
            }
            if (gelu) v = gelu_f(v);
            stage[rl * ELD + cl] = f2bf(v);
          }
        }
    }
    WT_STAMP(14);
    // (a wave only reads back its own region; the barrier orders the LDS traffic)
    if constexpr (PRE) { asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); __builtin_amdgcn_s_barrier(); }
    else __syncthreads();
    WT_STAMP(15);
    constexpr int CPR = WN / 8;          // 16-byte chunks per row
    constexpr int NIT = (WM * CPR + 63) / 64;
    bf16_t* Cb = (bf16_t*)p.C;
    const bf16_t* Rb = (const bf16_t*)p.residual;
    // residual chunks are fetched RB at a time ahead of their use (all loads of a batch back to back); RB = 4 keeps the 128 x 160
    // kernel at 2 waves per SIMD (10 chunks in flight at once cost 40 VGPRs and one of the two resident workgroups per CU)
    constexpr int RB = NIT < 4 ? NIT : 4;
    // RES: a residual is added.  PRE splits the two cases into separate bodies (a uniform branch): in ONE body the compiler waits
    // for the conditional residual loads unconditionally (vmcnt(0) at the join) and the residual-free GEMMs would drain the next
    // tile's DMA in front of their first C store.
    auto store_rows = [&](auto RESc) {
      constexpr bool RES = decltype(RESc)::value;
#pragma unroll
      for (int it0 = 0; it0 < NIT; it0 += RB) {
        uint4 rres[RB];
        if (RES && Rb) {
#pragma unroll
          for (int u = 0; u < RB; ++u) {
            const int idx = min((it0 + u) * 64 + lane, WM * CPR - 1);
            const int rl = idx / CPR, cch = idx - rl * CPR;
            const int row = min(mw + rl, p.M - 1), col = min(nw + cch * 8, p.N - 8);
            rres[u] = *(const uint4*)(Rb + (size_t)row * p.ldr + col);
          }
        }
#pragma unroll
        for (int u = 0; u < RB; ++u) {
          const int it = it0 + u;
          if (it >= NIT) break;
          const int idx = it * 64 + lane;
          const int rl = idx / CPR, cch = idx - rl * CPR;
          const int row = mw + rl, col = nw + cch * 8;
          if (idx < WM * CPR && row < p.M && col < p.N) {
            uint4 v = *(const uint4*)(stage + rl * ELD + cch * 8);
            if (RES && Rb) {
              float a[8], b[8];
              unpack8(v, a);
              unpack8(rres[u], b);
#pragma unroll
              for (int k = 0; k < 8; ++k) a[k] += b[k];
              v = pack8(a);
              if (p.colstats) *(uint4*)(stage + rl * ELD + cch * 8) = v;      // the statistics are those of the FINAL values
            }
            *(uint4*)(Cb + (size_t)row * p.ldc + col) = v;
          }
        }
      }
    };
    if constexpr (PRE) {
      if (Rb) store_rows(std::true_type{});
      else store_rows(std::false_type{});
    } else {
      store_rows(std::true_type{});
    }
    if (p.colstats) {
      // Per-column (sum, sum of squares) of this wave's output rows, one record per 32-row block: the GroupNorm that consumes
      // this tensor reduces these few floats instead of re-reading the whole activation (norm.hip, gn_finalize_cols_kernel).
      // The wave reads back its own staged (bf16, final) tile; LDS operations of one wave execute in order.
#pragma unroll
      for (int c0 = 0; c0 < WN; c0 += 64) {
        const int cl = c0 + lane, col = nw + cl;
        if (cl < WN && col < p.N) {
#pragma unroll
          for (int rb = 0; rb < WM / 32; ++rb) {
            if (mw + rb * 32 >= p.M) break;                               // M % 32 == 0 whenever statistics are requested
            float sm = 0.f, sq = 0.f;
#pragma unroll 8
            for (int r = 0; r < 32; ++r) {
              const float x = bf2f(stage[(rb * 32 + r) * ELD + cl]);
              sm += x; sq += x * x;
            }
            float* o = p.colstats + ((size_t)((mw >> 5) + rb) * p.N + col) * 2;
            o[0] = sm; o[1] = sq;
          }
        }
      }
    }
    return;
  }
  if (!partial && p.fast_f32) {
    // fp32 C with an optional fp32 residual, no accumulate (the fp32 residual stream of the CLIP-ViT: h = h + proj(...) stays fp32 as
    // under torch.autocast, [3P] open_clip ResidualAttentionBlock via encoder.py:153-154).  The accumulator layout already holds 32
    // consecutive columns of a row across lanes 0-31 (and of row + 4 across lanes 32-63): one store instruction writes two full
    // 128-byte lines.  All loads are unconditional (clamped) and issued per fragment before their first use.
    const float* Rf = (const float*)p.residual;
    float* Cf = (float*)p.C;
    const bool gelu = GENERAL && (p.flags & E4T_ACT_GELU) != 0;
#pragma unroll
    for (int i = 0; i < FM; ++i)
#pragma unroll
      for (int j = 0; j < FN; ++j) {
        const int col = nw + j * 32 + frow, colc = min(col, p.N - 1);
        const float bvv = p.bias ? p.bias[colc] : 0.f;
        float rr[16];
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const int row = min(mw + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * fhi, p.M - 1);
          rr[r] = Rf ? Rf[(size_t)row * p.ldr + colc] : 0.f;
        }
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const int row = mw + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * fhi;
          float v = acc[i][j][r] * p.alpha + bvv;
          if (gelu) v = gelu_f(v);
          v += rr[r];
          if (row < p.M && col < p.N) Cf[(size_t)row * p.ldc + col] = v;
        }
        __builtin_amdgcn_sched_barrier(0);
      }
    return;
  }
#pragma unroll
  for (int i = 0; i < FM; ++i)
#pragma unroll
    for (int j = 0; j < FN; ++j) {
      const int col = nw + j * 32 + frow;
      if (col >= p.N) continue;
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int row = mw + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * fhi;
        if (row >= p.M) continue;
        if (partial) p.ws[((size_t)(p.zslab >= 0 ? p.zslab : (int)blockIdx.z) * p.M + row) * p.N + col] = acc[i][j][r];  // z = batch*splitk + split
        else epilogue_store(p, acc[i][j][r], row, col);
      }
    }
}

// The same DMA through a buffer resource: 32-bit per-lane byte offset + wave-uniform SGPR byte offset; out-of-range offsets read
// as zero.  (Kept in a non-template helper: the builtin is not instantiable from a value-dependent context on the host pass.)
__device__ __forceinline__ void buf_dma16(__amdgpu_buffer_rsrc_t rs, unsigned voff, int soff, bf16_t* lds_wave_base) {
  __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, (__attribute__((address_space(3))) void*)lds_wave_base, 16, voff, soff, 0, 0);
}

// Workgroup -> output-tile mapping.  The dispatcher deals consecutive workgroups (x fastest) round-robin over the 8 XCDs,
// each with its own 4-MiB L2: with the plain blockIdx mapping the tiles sharing an A row panel (and, for the 3x3 convs,
// the neighbouring image rows of the halo) sit in 8 different L2s and every panel is fetched 8 times.  Re-deal so that
// the workgroups of one XCD own one CONTIGUOUS chunk of the tile raster, and walk that chunk in groups of 8 row panels
// so the ~64 tiles in flight on an XCD form a compact 8 x 8 block of the output.
__device__ __forceinline__ void xcd_tile(int& bx, int& by, int GM) {
  const int gx = gridDim.x, gy = gridDim.y, nwg = gx * gy;
  const int lin = blockIdx.y * gx + blockIdx.x;
  const int q = nwg >> 3, r = nwg & 7, v = lin & 7;
  const int lin2 = (v < r ? v * (q + 1) : r * (q + 1) + (v - r) * q) + (lin >> 3);
  const int per = GM * gx, grp = lin2 / per, l = lin2 - grp * per;
  const int first = grp * GM, gsz = min(gy - first, GM);
  bx = l / gsz;
  by = first + (l - bx * gsz);
}

// The same re-deal over a 3-D grid whose z index is a split-K slice (gemm_tn_kernel).  The dispatcher deals by LINEAR workgroup id
// (x fastest, then y, then z), so with the 2-D mapping above the tiles of one K-slice — which all stream the SAME rows of A and B,
// each reading its own 128-column stripe — sit in 8 different L2s and every 128-byte row segment is fetched by several of them:
// measured (profiles/r03_roofline_per_shape.csv) 3.2 x the algorithmic bytes on the K = 65536 weight gradients of the 64 x 64
// level, which at 6.8 TB/s of actual traffic are HBM-bound on exactly that waste.  Split-major chunks: one XCD owns whole slices.
__device__ __forceinline__ void xcd_tile3(int& bx, int& by, int& bz) {
  const int gx = gridDim.x, gy = gridDim.y, per = gx * gy, nwg = per * gridDim.z;
  const int lin = (blockIdx.z * gy + blockIdx.y) * gx + blockIdx.x;
  const int q = nwg >> 3, r = nwg & 7, v = lin & 7;
  const int lin2 = (v < r ? v * (q + 1) : r * (q + 1) + (v - r) * q) + (lin >> 3);
  bz = lin2 / per;
  const int t = lin2 - bz * per;
  by = t / gx;
  bx = t - by * gx;
}

// ---- channel-chunk-major K order for stride-1 3x3 convolutions (GemmArgs::chan_major) ---------------------------------------
// K = 9 taps x Cin.  Walked tap-major, every input pixel is read 9 times at a reuse distance of one whole tap panel per resident
// workgroup (256 rows x Cin x 2 B x 32 workgroups per XCD = 5-10 MB against a 4 MB L2): measured (profiles/r03_roofline_per_shape
// .csv) the 3x3 convs fetched 3.6-5.2 x their algorithmic bytes from beyond L2.  Walked channel-chunk-major — K-tile kt = (chunk
// kt / 9, tap kt % 9) — the 9 shifted reads of one chunk are adjacent in time (reuse distance: one K-tile per workgroup).  B is
// addressed by the same (tap, chunk), so the product only changes its fp32 summation order.  For stride 1 the tap shift is
// wave-uniform ((ky * Win + kx) * Cin elements from the pixel up-left of the output position) and goes into the SCALAR offset:
// the buffer base is lowered by one image row + one pixel (those bytes are never touched: lanes whose tap falls into the padding
// carry the out-of-range offset), the per-lane offset is the output position's own pixel for all 9 taps, and the only per-lane
// work per K-tile is picking that offset or an out-of-range one from a 9-bit validity mask.
// The walk is INCREMENTAL: every SIMD retires one instruction at a time whatever its kind (SQ counters, profiles/r03_gemm_pmc.txt:
// MFMA 32 cycles + ~4 per scalar / vector / LDS instruction add up to the kernel time), so decoding (chunk, tap) from the K-tile
// index with a division per issue (~20 SALU) cost the ping-pong kernels a scalar instruction per MFMA.  One walker per operand
// stream: tap and chunk offset live in SGPRs, the A-side tap shift comes from lane `tap` of a VGPR table (one v_readlane), the
// B-side one is tap * Cin * 2.
struct CmWalk {
  int tap, ch;                        // current tap (0..8); byte offset of the current channel chunk
  __device__ __forceinline__ void init(int kt, int ktw) { const int c = kt / 9; tap = kt - 9 * c; ch = c * ktw * 2; }
  __device__ __forceinline__ void next(int ktw) {
    ++tap;
    if (tap == 9) { tap = 0; ch += ktw * 2; }
  }
};
// lane t < 9 holds the A-side byte shift of tap t
__device__ __forceinline__ int cm_tap_table(const GemmArgs& p, int lane) {
  const int t = lane < 9 ? lane : 0;
  return ((t / 3) * p.Win + (t % 3)) * p.Cin * 2;
}
// (readfirstlane: both offsets are wave-uniform by construction; saying so keeps them in SGPRs — where the compiler had moved
// the B-side multiply to the VALU it wrapped every buffer_load ... lds in a waterfall loop over the "divergent" scalar offset)
__device__ __forceinline__ int cm_a_so(int table, const CmWalk& w) { return __builtin_amdgcn_readfirstlane(__builtin_amdgcn_readlane(table, w.tap) + w.ch); }
__device__ __forceinline__ int cm_b_so(const GemmArgs& p, const CmWalk& w) { return __builtin_amdgcn_readfirstlane(w.tap * (p.Cin * 2) + w.ch); }
// per-lane offset of one row for the walker's tap: the row's own pixel, or all-ones (out of range) when the INVERTED validity
// mask has the tap's bit set — two VALU instructions (v_bfe_i32 with the scalar tap, v_or)
__device__ __forceinline__ unsigned cm_row_off(unsigned center, int inv_mask, const CmWalk& w) {
  return center | (unsigned)__builtin_amdgcn_sbfe(inv_mask, (unsigned)w.tap, 1u);
}
__device__ __forceinline__ int cm_inv_mask(const GemmArgs& p, bool ok, int oy, int ox) {
  int m = 0;
#pragma unroll
  for (int ky = 0; ky < 3; ++ky)
#pragma unroll
    for (int kx = 0; kx < 3; ++kx) {
      const int iy = oy + ky - 1, ix = ox + kx - 1;
      if (ok && iy >= 0 && iy < p.Hin && ix >= 0 && ix < p.Win) m |= 1 << (ky * 3 + kx);
    }
  return ~m;                          // inverted: bit t set = tap t of this row reads padding (cm_row_off)
}
__device__ __forceinline__ unsigned cm_center(const GemmArgs& p, long long img_px, int oy, int ox, int kc) {
  return (unsigned)(((img_px + (long long)oy * p.Win + ox) * p.Cin + kc) * 2);
}
__device__ __forceinline__ unsigned cm_shift_bytes(const GemmArgs& p) { return (unsigned)((p.Win + 1) * p.Cin * 2); }
// (every input of the descriptor goes through readfirstlane: all of them ARE wave-uniform, but a descriptor the compiler cannot PROVE
// uniform is kept in VGPRs and every buffer_load ... lds that uses it is wrapped in a waterfall loop — 4 v_readfirstlane + compare +
// s_and_saveexec + loop per DMA instruction, seen when the channel-major walk became a template parameter in round 5)
__device__ __forceinline__ __amdgpu_buffer_rsrc_t uniform_rsrc(const void* base, unsigned bytes) {
  const unsigned long long b = (unsigned long long)base;
  const unsigned lo = __builtin_amdgcn_readfirstlane((unsigned)b), hi = __builtin_amdgcn_readfirstlane((unsigned)(b >> 32));
  return __builtin_amdgcn_make_buffer_rsrc((void*)(((unsigned long long)hi << 32) | lo), 0, (int)__builtin_amdgcn_readfirstlane(bytes), 0x00020000);
}
__device__ __forceinline__ __amdgpu_buffer_rsrc_t cm_rsrc(const GemmArgs& p) {
  const unsigned sh = cm_shift_bytes(p);
  return uniform_rsrc((const char*)p.A - sh, p.a_bytes + sh);
}

// ---- tail rows of a dense GEMM -------------------------------------------------------------------------------------------------
// The CLIP-ViT's token matrix at B = 16 has 16 x 257 = 4112 = 32 x 128 + 16 rows.  Every tiling of 4112 rows pays a whole extra
// round of workgroups for the last 16: measured in the step (profiles/r04_prefetch_off_roofline_per_shape.csv), the SAME N / K at
// M = 4096 (the UNet's 16 x 16 level) and at M = 4112 take 24.7 vs 44.9 us (N1280 K1280), 73.6 vs 89.6 + 35.9 us (N1280 K5120: the
// 4112-row plan needs split-K and its reduce), 58.7 vs 94.6 us (N5120 K1280).  The launcher therefore plans such a GEMM for its
// first M - r rows (r = M % 128 <= 32) and hands the r tail rows to this function, which runs at the END of the same launch in the
// workgroups with the lowest linear ids: one 32-column block of the tail per workgroup, the K range split evenly over the workgroup's
// waves, operands read straight from global memory into MFMA fragments (16 bytes per lane and k-step for each operand: the
// fragment layout of v_mfma_f32_32x32x16_bf16 is 8 consecutive k per lane, no LDS needed), the waves' partial accumulators summed
// through LDS in wave order (deterministic), then the ordinary scalar epilogue.  A 16 x N x K tail is 0.4 % of the GEMM's work.
// Requires: dense A (no second source, no panels), no split-K / batch, K % 16 == 0, 16-byte aligned rows (the launcher checks).
// `red`: NW x 16 x 64 floats of LDS (the kernel's operand buffers: dead after the main epilogue).
// (GELU only in the GENERAL instantiations, as everywhere else: its expansion next to a plain epilogue costs registers)
template <int NW, bool GENERAL>
__device__ __forceinline__ void gemm_tail(const GemmArgs& p, float* red, int wave_v, int lane) {
  const int wave = __builtin_amdgcn_readfirstlane(wave_v);   // (uniform by construction: scalar loop control)
  const int nblk = (p.N + 31) >> 5;
  const int lin = blockIdx.y * gridDim.x + blockIdx.x, nwg = gridDim.x * gridDim.y;
  if (lin >= nblk) return;                                   // workgroup-uniform
  const int ksteps = p.K >> 4;
  const int per = (ksteps + NW - 1) / NW;
  const int ks0 = wave * per, ks1 = min(ks0 + per, ksteps);
  const int frow = lane & 31, fhi = lane >> 5;
  // rows >= tail_rows of the 32-row fragment re-read the last valid row: row i of C depends on row i of A only, and those rows are never stored
  const bf16_t* Ap = p.A + (size_t)(p.tail_row0 + min(frow, p.tail_rows - 1)) * p.lda + fhi * 8;
  for (int blk = lin; blk < nblk; blk += nwg) {
    const int n0 = blk << 5;
    const bf16_t* Bp = p.B + (size_t)min(n0 + frow, p.N - 1) * p.ldb + fhi * 8;
    f32x16 acc;
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[r] = 0.f;
    int ks = ks0;
    for (; ks + 8 <= ks1; ks += 8) {                         // 16 independent 16-byte loads in flight per lane
      bf16x8 a[8], b[8];
#pragma unroll
      for (int u = 0; u < 8; ++u) {
        a[u] = *(const bf16x8*)(Ap + (ks + u) * 16);
        b[u] = *(const bf16x8*)(Bp + (ks + u) * 16);
      }
      __builtin_amdgcn_sched_barrier(0);                     // all 16 loads leave before the first MFMA waits for one (the scheduler rolled them into a 6-deep window)
#pragma unroll
      for (int u = 0; u < 8; ++u) acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[u], b[u], acc, 0, 0, 0);
    }
    for (; ks < ks1; ++ks)
      acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(*(const bf16x8*)(Ap + ks * 16), *(const bf16x8*)(Bp + ks * 16), acc, 0, 0, 0);
    __syncthreads();                                         // the LDS is free: main epilogue / previous block's reduction done
#pragma unroll
    for (int r = 0; r < 16; ++r) red[(wave * 16 + r) * 64 + lane] = acc[r];
    __syncthreads();
    if (wave == 0) {
      const int col = n0 + frow;
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        float v = 0.f;
#pragma unroll
        for (int w = 0; w < NW; ++w) v += red[(w * 16 + r) * 64 + lane];
        const int row = (r & 3) + 8 * (r >> 2) + 4 * fhi;    // C/D layout of the 32 x 32 MFMA
        if (row < p.tail_rows && col < p.N) epilogue_store<GENERAL>(p, v, p.tail_row0 + row, col);
      }
    }
  }
}

template <int N>
__device__ __forceinline__ void wait_vmcnt() { asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory"); }

// The K-loop barrier of the DMA kernels.  After it, some wave starts the LDS-DMA of a new tile INTO THE STAGE EVERY WAVE READ IN
// THE ITERATION BEFORE, so every wave's fragment reads of that stage must have COMPLETED, not merely been issued, when it
// arrives.  s_barrier alone does not order that: the compiler sinks the last k-step's MFMA — and the lgkmcnt wait in front of
// it — below the barrier (gfx950 needs no counter drain at s_barrier), leaving ds_reads in flight across it.  Measured: with
// three 4-wave workgroups per CU (64 x 64 tile, 3 stages) 0-2 of 1280 output tiles per launch came out wrong, not
// reproducibly — a DMA that hit in L2 landed before a ds_read queued behind the other workgroups' LDS traffic had executed.
// Draining lgkmcnt first closes the window for every stage count.
__device__ __forceinline__ void loop_barrier() {
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  __builtin_amdgcn_s_barrier();
}

}  // namespace
