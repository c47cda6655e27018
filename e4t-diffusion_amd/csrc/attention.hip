// Fused multi-head attention  O = softmax(Q K^T * scale) V  — forward and backward — for gfx950.
//
// Replaces the reference's SDPA / baddbmm+softmax+bmm / xFormers processors
// (e4t/models/cross_attention.py:521-531, 222-251 + 313-314, 473-481) and the open_clip ViT's
// nn.MultiheadAttention core.  No (T x S) score matrix is ever written to HBM.
//
// Layout: Q, K, V, O, dO, dQ, dK, dV are the projection GEMMs' outputs as they are:
// row-major (B*T, heads*DH) bf16 with an arbitrary row stride; head h is the column slice
// [h*DH, (h+1)*DH).  No head split / merge permutes exist anywhere.
//
// MFMA mapping (v_mfma_f32_32x32x16_bf16, K = 16 so DH = 40/80/160/64 pad to 48/80/160/64 only):
//   forward, one wave = 32 queries:   S^T[key][q] = K . Q^T   (A = K tile from LDS, B = Q in registers)
//     -> every lane owns ONE query column (q = lane & 31) and 16 of the 32 keys in registers, so the
//        row max / row sum are 16 in-register ops + one cross-half exchange, and m, l are per-lane scalars;
//     O^T[d][q] += V^T . P^T   (A = V^T gathered from the ROW-MAJOR V tile in LDS with the transposing read
//        ds_read_b64_tr_b16, B = P^T straight from the S^T accumulator registers)
//     -> the S^T accumulator layout IS the B-operand layout once the 32 keys of a sub-tile are
//        enumerated in the order the accumulator holds them ({4hi..4hi+3} u {4hi+8..4hi+11} per k-step): two
//        4-row gathers supply V^T in exactly that order, so P never goes through LDS or a cross-lane shuffle
//        and no transposed image of V is ever written.  O^T keeps q in the lane: rescale is a scalar.
//   backward dQ: same skeleton (dQ^T[d][q] += K^T . dS^T).
//   backward dK/dV, one wave = 32 keys: S[q][key] = Q . K^T (key in the lane), dV^T[d][key] += dO^T . P,
//     dK^T[d][key] += Q^T . dS.   Two kernels (dQ | dK,dV) -> no atomics, bitwise deterministic.
// Softmax statistics are fp32, exp2 domain (scale * log2 e folded in).  LSE is stored in log2 units.
#include "common.h"
#include "../../include/e4t_hip.h"
#include <math.h>
#include <type_traits>

namespace {

#ifndef DKV_WAVES
#define DKV_WAVES 2
#endif

struct AttnArgs {
  const bf16_t *Q, *K, *V, *dO;
  const bf16_t* O;
  bf16_t *Out, *dQ, *dK, *dV;
  float* L;      // [B][H][T] log2-domain logsumexp
  float* Delta;  // [B][H][T] rowsum(dO * O)
  float* LD;     // [B][H][T][2] {L, Delta} pairs (written by the dQ kernel next to Delta for the DMA-staged dK/dV kernel) or null
  int T, S, H;
  int ldq, ldk, ldv, ldo;      // row strides (elements)
  long long bq, bk, bv, bo;    // batch strides (elements)
  float scale2;                // scale * log2(e)
  float scale;
  int causal;                  // key index > query index is masked
  int xcd_raster;              // re-deal workgroups so one XCD owns whole (batch, head) pairs (xcd_block)
  // dK/dV kernel with few keys (cross-attention, S = 77): the query range is cut into `tsplit` chunks of `tchunk` queries, one
  // workgroup each, writing fp32 partials part[tsplit][B][H][2 (dV, dK)][S][DH]; attn_dkv_reduce_kernel sums them in split order
  float* part;
  int tsplit, tchunk;
};

// Workgroups are dealt to the 8 XCDs round-robin by linear id (x fastest), so the row blocks of one (batch, head) — which all
// stream the SAME K and V (dK/dV kernel: the same Q and dO) — land in 8 different L2s and every operand is fetched 8 times from
// HBM/MALL (measured r03: 3.5-4.5 x the algorithmic bytes per launch at T = 4096).  Re-deal so that each XCD owns a contiguous
// chunk of the (block, head, batch) raster: its ~128 resident workgroups then cover ~4 whole heads whose K/V (0.66 MB a head at
// T = 4096, dh = 40) stay in its 4 MB L2.
struct Blk { int x, h, b; };
__device__ __forceinline__ Blk xcd_block(int on) {
  Blk o{(int)blockIdx.x, (int)blockIdx.y, (int)blockIdx.z};
  if (!on) return o;
  const int gx = gridDim.x, gy = gridDim.y, nwg = gx * gy * gridDim.z;
  const int lin = (blockIdx.z * gy + blockIdx.y) * gx + blockIdx.x;
  const int q = nwg >> 3, r = nwg & 7, v = lin & 7;
  const int lin2 = (v < r ? v * (q + 1) : r * (q + 1) + (v - r) * q) + (lin >> 3);
  const int t = lin2 / gx;
  o.x = lin2 - t * gx; o.b = t / gy; o.h = t - o.b * gy;
  return o;
}

template <int DH>
struct Cfg {
  static constexpr int DK = (DH + 15) / 16 * 16;  // contraction padding for Q.K^T
  static constexpr int DV = (DH + 31) / 32 * 32;  // row-tile padding for the transposed accumulators
  static constexpr int NKS = DK / 16;
  static constexpr int NDT = DV / 32;
  // LDS images are row-major [64 rows][LDE] with 256-B (DH = 160: 512-B) rows; the 16-B slot L of row r lives at slot
  // (((L>>2) ^ (r&3)) << 2) | ((L&3) ^ ((r>>2)&3)): conflict-free both for the row-major ds_read_b128 fragments (lane
  // groups {0-3,12-15,20-27} / {4-11,16-19,28-31}) and for the transposing 4-row x 64-B gathers (no single padded stride is).
  static constexpr int LDE = DH <= 128 ? 128 : 256;
  static constexpr int NCH = DH / 8;   // 16-byte chunks per source row
  static constexpr int NIT = (32 * NCH + 255) / 256;   // (row pair, chunk) items per thread per 64-row tile
};

union Frag {
  bf16x8 v;
  uint4 q;
  uint2 h[2];
  uint32_t w[4];
};

// max / sum of a value with its partner lane in the other half of the wave (lane ^ 32) as ONE VALU lane swap (gfx950
// v_permlane32_swap) instead of a ds_bpermute round trip through the LDS crossbar: the softmax row max sits at the head of the
// per-sub-tile dependency chain, and a bpermute there also drains every LDS read issued before it (lgkmcnt is in-order), which
// would defeat the fragment prefetch below.  (The clang builtin folds away its second result, hence inline asm; the s_nops
// cover the VALU-write -> permlane-read and permlane-write -> VALU-read hazards the compiler cannot see inside the asm.)
#ifndef ATTN_FWD_PREFETCH
#define ATTN_FWD_PREFETCH 0      // measured neutral (667 -> 658 us at dh 40): the SIMD serialises instruction cycles, not latency
#endif
#ifndef ATTN_PERMLANE
#define ATTN_PERMLANE 1
#endif
__device__ __forceinline__ float xhalf_max(float v) {
  if (!ATTN_PERMLANE) return fmaxf(v, __shfl_xor(v, 32, 64));
  float a = v, b = v;
  asm volatile("s_nop 1\n\tv_permlane32_swap_b32 %0, %1\n\ts_nop 1\n\tv_max_f32 %0, %0, %1" : "+v"(a), "+v"(b));
  return a;
}
__device__ __forceinline__ float xhalf_sum(float v) {
  if (!ATTN_PERMLANE) return v + __shfl_xor(v, 32, 64);
  float a = v, b = v;
  asm volatile("s_nop 1\n\tv_permlane32_swap_b32 %0, %1\n\ts_nop 1\n\tv_add_f32 %0, %0, %1" : "+v"(a), "+v"(b));
  return a;
}

// tools/probe/attn_probe.hip compiles timing variants of the forward kernel (results are wrong in every variant but 0):
//   1 v_exp_f32 replaced by the plain fma   2 no softmax VALU at all   3 MFMAs replaced by one VALU add   4 no tile staging
//   5 no LDS fragment reads (operands made up in registers)   6 = 4 + 5   7 = 5 + 2   8 = 5 + 3
#ifndef ATTN_PROBE
#define ATTN_PROBE 0
#endif
__device__ __forceinline__ f32x16 probe_mfma(bf16x8 a, bf16x8 b, f32x16 c) {
  if (ATTN_PROBE == 3 || ATTN_PROBE == 8) {
    Frag fa, fb;
    fa.v = a; fb.v = b;
    c[0] += __builtin_bit_cast(float, fa.w[0] ^ fb.w[1]);
    return c;
  }
  return __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c, 0, 0, 0);
}


__device__ __forceinline__ float fast_exp2(float x) { return __builtin_amdgcn_exp2f(x); }

// The softmax arithmetic of two adjacent score elements per instruction (v_pk_fma_f32 / v_pk_add_f32 / v_pk_mul_f32): the kernels
// are bound by VALU issue slots, and a wave64 VALU instruction costs its SIMD ~4.5 clk packed or not (tools/probe/mfma_rate.hip).
// Same IEEE operations as the scalar form (s * scale2 - m contracts to one fma either way): results are bit-identical.
#ifndef ATTN_DKV_FOLD
#define ATTN_DKV_FOLD 1
#endif
#ifndef ATTN_PACKED
#define ATTN_PACKED 1
#endif
typedef float f32x2 __attribute__((ext_vector_type(2)));
// Packed fp32 VALU (v_pk_fma_f32 / v_pk_mul_f32) beside MFMAs: measured per kernel (round 6, tools/ab_attn_all.py, one box).  In the
// DMA-staged dh-40 kernels it costs MORE than the two scalar ops it replaces (forward 471 -> 445 us, backward 1564 -> 1533 us); the
// register-staged kernels (3 waves per SIMD at dh <= 64, the dh-80 / 160 ones) run 2-3 % faster WITH it.  So the pair helpers take the
// choice as a template argument (ATTN_PK_OLD / ATTN_PK_DMA, 0 / 1, for the A/B), and the file is compiled with
// -fno-slp-vectorize so that the compiler packs nothing on its own.
#ifndef ATTN_PK_OLD
#define ATTN_PK_OLD 1
#endif
#ifndef ATTN_PK_DMA
#define ATTN_PK_DMA 0
#endif
template <bool PK>
__device__ __forceinline__ f32x2 pk_fma(f32x2 a, f32x2 b, f32x2 c) {
  if constexpr (PK) return __builtin_elementwise_fma(a, b, c);
  else return f32x2{__builtin_fmaf(a.x, b.x, c.x), __builtin_fmaf(a.y, b.y, c.y)};
}
template <bool PK>
__device__ __forceinline__ f32x2 pk_mul(f32x2 a, f32x2 b) {
  if constexpr (PK) return a * b;
  else return f32x2{a.x * b.x, a.y * b.y};
}


// element offset of the logical 16-byte slot L of tile row r in a swizzled LDS image (see Cfg)
template <int LDE>
__device__ __forceinline__ int img_off(int r, int L) {
  return r * LDE + (((((L >> 2) ^ (r & 3)) << 2) | ((L & 3) ^ ((r >> 2) & 3))) << 3);
}

// One 64-row x DH tile of a row-major global matrix, held in registers between its global load and its LDS
// stores so that the load of tile t+1 is in flight while tile t is being consumed.  Item = (row pair, 16-B chunk),
// lanes running along the row pairs: the 32 lanes of a ds_write_b32 group of the transposed store then hit 32
// different columns of ONE d row = 32 different banks (chunk-major lanes were 5-way conflicted at dh=40: rows 8 apart
// are 288 dwords apart = the same bank; PMC: a third of all LDS cycles were those conflicts).
template <int DH>
struct TileRegs {
  using C = Cfg<DH>;
  uint4 v[C::NIT][2];
  unsigned vo[C::NIT];          // byte offset of this thread's (row pair, chunk) inside a 64-row tile; out of range for idle threads
  int ldb;                      // row stride in bytes
  // Tiles are fetched through a buffer resource whose extent ends with the last valid row: rows beyond it (ragged last tile)
  // and idle threads read zeros from the hardware, and stepping to the next tile is one scalar add — no per-tile address VALU
  // on kernels that are VALU-bound.
  __device__ __forceinline__ void init(int ld) {
    ldb = ld * 2;
#pragma unroll
    for (int it = 0; it < C::NIT; ++it) {
      const int item = threadIdx.x + it * 256;
      const int ch = item >> 5, kp = item & 31;
      vo[it] = item < 32 * C::NCH ? (unsigned)((2 * kp * ld + ch * 8) * 2) : 0xFFFF0000u;
    }
  }
  __device__ __forceinline__ void load(__amdgpu_buffer_rsrc_t rs, int r0) {
    const int soff = r0 * ldb;
#pragma unroll
    for (int it = 0; it < C::NIT; ++it) {
      v[it][0] = buf_load16(rs, vo[it], soff);
      v[it][1] = buf_load16(rs, vo[it], soff + ldb);
    }
  }
  // swizzled row-major image lds[64][LDE]
  __device__ __forceinline__ void store_rows(bf16_t* lds) const {
#pragma unroll
    for (int it = 0; it < C::NIT; ++it) {
      const int item = threadIdx.x + it * 256;
      const int ch = item >> 5, kp = item & 31;
      if (item < 32 * C::NCH) {
        *(uint4*)(lds + img_off<C::LDE>(2 * kp, ch)) = v[it][0];
        *(uint4*)(lds + img_off<C::LDE>(2 * kp + 1, ch)) = v[it][1];
      }
    }
  }
};

// the zero pad slots [DH, DK) of an LDS image are written once (the tile stores never touch them)
template <int DH>
__device__ __forceinline__ void zero_pad_cols(bf16_t* lds) {
  constexpr int PADC = (Cfg<DH>::DK - DH) / 8;
  if (PADC > 0)
    for (int i = threadIdx.x; i < 64 * PADC; i += 256)
      *(uint4*)(lds + img_off<Cfg<DH>::LDE>(i / PADC, Cfg<DH>::NCH + i % PADC)) = make_uint4(0, 0, 0, 0);
}

// B-operand fragments of a row held in registers: 8 consecutive d starting at ks*16 + hi*8 (zero beyond DH / R)
template <int DH>
__device__ __forceinline__ void load_row_frags(const bf16_t* g, int ld, int row, int R, int hi, bf16x8* f) {
#pragma unroll
  for (int ks = 0; ks < Cfg<DH>::NKS; ++ks) {
    Frag t;
    t.q = make_uint4(0, 0, 0, 0);
    const int d0 = ks * 16 + hi * 8;
    if (row < R && d0 < DH) t.q = *(const uint4*)(g + (long long)row * ld + d0);
    f[ks] = t.v;
  }
}

// Per-lane fragment offsets into a swizzled image (elements; add (32 sub + 16 k2) * LDE for the sub-tile / k-step):
//   row[ks]      row-major A operand: tile row = lane & 31, the 16-B slot of MFMA k-step ks
//   tr_lo/hi[dt] transposed A operand (rows d = 32 dt + (lane & 31), 8 k-slots = tile rows {4hi..4hi+3} u {4hi+8..4hi+11}):
//                source address of this lane for the two ds_read_b64_tr_b16 gathers (source lane i' of a 16-lane group
//                supplies 4 contiguous elements of row (i' >> 2) at column 16 (group & 1) + 4 (i' & 3); output lane i
//                receives column i of those 4 rows — probed on hardware, tools/probe/tr_probe.hip)
template <int DH>
struct FragOff {
  using C = Cfg<DH>;
  int row[C::NKS], tr_lo[C::NDT], tr_hi[C::NDT];
  __device__ __forceinline__ FragOff() {
    const int lane = threadIdx.x & 63, li = lane & 31, hi = lane >> 5, il = lane & 15, d16 = (lane >> 4) & 1;
#pragma unroll
    for (int ks = 0; ks < C::NKS; ++ks) row[ks] = img_off<C::LDE>(li, ks * 2 + hi);
#pragma unroll
    for (int dt = 0; dt < C::NDT; ++dt) {
      const int L = dt * 4 + 2 * d16 + ((il & 3) >> 1), r = 4 * hi + (il >> 2);
      tr_lo[dt] = img_off<C::LDE>(r, L) + 4 * (il & 1);
      tr_hi[dt] = img_off<C::LDE>(r + 8, L) + 4 * (il & 1);
    }
  }
};
template <int DH>
__device__ __forceinline__ bf16x8 load_row_frag(const bf16_t* img, const FragOff<DH>& f, int sub, int ks) {
  if (ATTN_PROBE >= 5) { Frag t; t.w[0] = t.w[1] = t.w[2] = t.w[3] = 0x3C003C00u + threadIdx.x + sub + ks; return t.v; }
  return *(const bf16x8*)(img + sub * 32 * Cfg<DH>::LDE + f.row[ks]);
}
template <int DH>
__device__ __forceinline__ bf16x8 load_T_frag(const bf16_t* img, const FragOff<DH>& f, int dt, int sub, int k2) {
  if (ATTN_PROBE >= 5) { Frag t; t.w[0] = t.w[1] = t.w[2] = t.w[3] = 0x3C003C00u + threadIdx.x + sub + k2 + dt; return t.v; }
  const bf16_t* b = img + (sub * 32 + k2 * 16) * Cfg<DH>::LDE;
  return tr_frag(b + f.tr_lo[dt], b + f.tr_hi[dt]);
}
// accumulator registers [8*k2, 8*k2+8) -> bf16 B-operand
__device__ __forceinline__ bf16x8 pack_acc(const float* p, int k2) {
  Frag t;
#pragma unroll
  for (int j = 0; j < 4; ++j) t.w[j] = pack2bf(p[8 * k2 + 2 * j], p[8 * k2 + 2 * j + 1]);
  return t.v;
}
__device__ __forceinline__ int acc_row(int r, int hi) { return (r & 3) + 8 * (r >> 2) + 4 * hi; }

// transposed accumulator [d][row] -> global[row][d] (bf16), 4 consecutive d per 8-byte store
template <int DH>
__device__ __forceinline__ void store_T_acc(const f32x16* acc, float mul, bf16_t* g, int ld, int row, int R, int hi) {
  if (row >= R) return;
#pragma unroll
  for (int dt = 0; dt < Cfg<DH>::NDT; ++dt)
#pragma unroll
    for (int rg = 0; rg < 4; ++rg) {
      const int d = dt * 32 + 8 * rg + 4 * hi;
      if (d < DH) {
        float f[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) f[j] = acc[dt][rg * 4 + j] * mul;
        *(uint2*)(g + (long long)row * ld + d) = pack4(f);
      }
    }
}

// the same accumulator as fp32 rows of a dense [R][DH] partial (16-byte stores)
template <int DH>
__device__ __forceinline__ void store_T_acc_f32(const f32x16* acc, float* g, int row, int R, int hi) {
  if (row >= R) return;
#pragma unroll
  for (int dt = 0; dt < Cfg<DH>::NDT; ++dt)
#pragma unroll
    for (int rg = 0; rg < 4; ++rg) {
      const int d = dt * 32 + 8 * rg + 4 * hi;
      if (d < DH) *(float4*)(g + (size_t)row * DH + d) = make_float4(acc[dt][rg * 4], acc[dt][rg * 4 + 1], acc[dt][rg * 4 + 2], acc[dt][rg * 4 + 3]);
    }
}

// ================================================================================================
// forward
// ================================================================================================
template <int DH>
__global__ __launch_bounds__(256, (DH <= 64 ? 4 : DH <= 80 ? 2 : 1)) void attn_fwd_kernel(AttnArgs p) {
  using C = Cfg<DH>;
  __shared__ __attribute__((aligned(16))) bf16_t Ks[64 * C::LDE];
  __shared__ __attribute__((aligned(16))) bf16_t Vs[64 * C::LDE];
  const FragOff<DH> fo;
  const Blk blk = xcd_block(p.xcd_raster);
  const int b = blk.b, h = blk.h;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int li = lane & 31, hi = lane >> 5;
  const int q = blk.x * 128 + wave * 32 + li;
  const bf16_t* Qb = p.Q + b * p.bq + h * DH;
  const bf16_t* Kb = p.K + b * p.bk + h * DH;
  const bf16_t* Vb = p.V + b * p.bv + h * DH;

  bf16x8 qf[C::NKS];
  load_row_frags<DH>(Qb, p.ldq, q, p.T, hi, qf);

  f32x16 o[C::NDT];
#pragma unroll
  for (int dt = 0; dt < C::NDT; ++dt)
#pragma unroll
    for (int r = 0; r < 16; ++r) o[dt][r] = 0.f;
  float m = -INFINITY, l = 0.f;
  constexpr bool PREFETCH = DH <= 48 && ATTN_FWD_PREFETCH;      // (dh 64 would spill at four workgroups per CU)
  constexpr bool SUM_BY_MFMA = C::DV > DH && DH % 8 == 0;      // a spare O^T row (d = DH) exists: it accumulates sum_k p

  TileRegs<DH> kr, vr;
  const __amdgpu_buffer_rsrc_t rsK = make_rsrc(Kb, (unsigned)(((long long)(p.S - 1) * p.ldk + DH) * 2));
  const __amdgpu_buffer_rsrc_t rsV = make_rsrc(Vb, (unsigned)(((long long)(p.S - 1) * p.ldv + DH) * 2));
  kr.init(p.ldk); vr.init(p.ldv);
  kr.load(rsK, 0);
  vr.load(rsV, 0);
  zero_pad_cols<DH>(Ks);
  if (SUM_BY_MFMA) {       // V image: element d = DH of every key row = 1.0 (bf16), the rest of that 16-B slot = 0; never overwritten
    if (threadIdx.x < 64) *(uint4*)(Vs + img_off<C::LDE>(threadIdx.x, C::NCH)) = make_uint4(0x3F80u, 0, 0, 0);
  }
  for (int kv0 = 0; kv0 < p.S; kv0 += 64) {
    if ((ATTN_PROBE != 4 && ATTN_PROBE != 6) || kv0 == 0) {
    __syncthreads();                       // everyone finished reading the previous tile
    kr.store_rows(Ks);
    vr.store_rows(Vs);
    __syncthreads();
    if (kv0 + 64 < p.S) {                  // next tile's loads fly under this tile's MFMAs
      kr.load(rsK, kv0 + 64);
      vr.load(rsV, kv0 + 64);
    }
    }
    const int nsub = (p.S - kv0 > 32) ? 2 : 1;
    // PREFETCH (dh <= 48): every LDS fragment is requested one phase before the MFMA that consumes it — this sub-tile's V^T
    // fragments and the next sub-tile's K fragments are issued in front of the softmax VALU block and land underneath it
    // (measured with tools/probe/attn_probe.hip: with the fragment reads taken out the dh = 40 forward drops 667 -> 464 us, i.e.
    // a third of the kernel was exposed LDS latency — the compiler had sunk each ds_read to just above its MFMA).
    bf16x8 kfr[C::NKS];
    if (PREFETCH) {
#pragma unroll
      for (int ks = 0; ks < C::NKS; ++ks) kfr[ks] = load_row_frag<DH>(Ks, fo, 0, ks);
    }
#pragma unroll
    for (int sub = 0; sub < 2; ++sub) {
      if (sub >= nsub) break;
      f32x16 s;
#pragma unroll
      for (int r = 0; r < 16; ++r) s[r] = 0.f;
#pragma unroll
      for (int ks = 0; ks < C::NKS; ++ks) {
        s = probe_mfma(PREFETCH ? kfr[ks] : load_row_frag<DH>(Ks, fo, sub, ks), qf[ks], s);
      }
      bf16x8 vfr[C::NDT][2];
      if (PREFETCH) {
#pragma unroll
        for (int k2 = 0; k2 < 2; ++k2)
#pragma unroll
          for (int dt = 0; dt < C::NDT; ++dt) vfr[dt][k2] = load_T_frag<DH>(Vs, fo, dt, sub, k2);
        if (sub == 0 && nsub > 1) {
#pragma unroll
          for (int ks = 0; ks < C::NKS; ++ks) kfr[ks] = load_row_frag<DH>(Ks, fo, 1, ks);
        }
        __builtin_amdgcn_sched_barrier(0);
      }
      // Softmax bookkeeping is VALU work on a VALU-bound kernel (PMC: ~18 VALU instructions per MFMA), so it is kept minimal:
      //  * the row max is taken on the raw scores and the scale folded into the exp argument (one fma per element);
      //  * the running max m is only raised when some row of the wave exceeds it by more than 2^8 ("lazy rescaling"):
      //    p = exp2(s - m) <= 256 stays harmless in fp32 / bf16, and the O / l rescale (alpha) is skipped on most tiles;
      //    m, l and LSE = m + log2 l stay mutually consistent, so the result is unchanged;
      //  * for head dims with a spare accumulator row (40, 80) the row sum l is produced by the P.V MFMA itself (V image
      //    column DH holds ones), not by 16 adds + a cross-half exchange per sub-tile.
      float pr[16];
      float mx = -INFINITY;
      const bool tail = kv0 + 64 > p.S;        // ragged last tile only: keys beyond S get -inf
      if (tail || p.causal) {
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const int key = kv0 + sub * 32 + acc_row(r, hi);
          s[r] = (key < p.S && (!p.causal || key <= q)) ? s[r] : -INFINITY;
        }
      }
      if (ATTN_PROBE != 2 && ATTN_PROBE != 7) {
#pragma unroll
      for (int r = 0; r < 16; ++r) mx = fmaxf(mx, s[r]);
      mx = xhalf_max(mx) * p.scale2;         // scale2 > 0
      }
      if (ATTN_PROBE != 2 && ATTN_PROBE != 7 && __builtin_amdgcn_ballot_w64(mx > m + 8.f) != 0) {
        const float mn = fmaxf(m, mx);
        const float alpha = fast_exp2(m - mn);
        m = mn;
        if (!SUM_BY_MFMA) l *= alpha;
#pragma unroll
        for (int dt = 0; dt < C::NDT; ++dt)
#pragma unroll
          for (int r = 0; r < 16; ++r) o[dt][r] *= alpha;
      }
#pragma unroll
      for (int r = 0; r < 16; r += 2) {
        if (ATTN_PACKED && ATTN_PROBE == 0) {
          const f32x2 t = pk_fma<ATTN_PK_OLD>(f32x2{s[r], s[r + 1]}, f32x2{p.scale2, p.scale2}, f32x2{-m, -m});
          pr[r] = fast_exp2(t.x); pr[r + 1] = fast_exp2(t.y);
        } else {
#pragma unroll
          for (int e = r; e < r + 2; ++e) pr[e] = (ATTN_PROBE == 2 || ATTN_PROBE == 7) ? s[e] : ATTN_PROBE == 1 ? s[e] * p.scale2 - m : fast_exp2(s[e] * p.scale2 - m);
        }
      }
      if (!SUM_BY_MFMA) {
        float rs = 0.f;
#pragma unroll
        for (int r = 0; r < 16; ++r) rs += pr[r];
        l += xhalf_sum(rs);
      }
      const bf16x8 pf0 = pack_acc(pr, 0), pf1 = pack_acc(pr, 1);
      if (PREFETCH) {
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int k2 = 0; k2 < 2; ++k2)             // k-step outer: neighbouring MFMAs accumulate into different O^T blocks
#pragma unroll
          for (int dt = 0; dt < C::NDT; ++dt) o[dt] = probe_mfma(vfr[dt][k2], k2 ? pf1 : pf0, o[dt]);
      } else {
#pragma unroll
        for (int dt = 0; dt < C::NDT; ++dt) {
          o[dt] = probe_mfma(load_T_frag<DH>(Vs, fo, dt, sub, 0), pf0, o[dt]);
          o[dt] = probe_mfma(load_T_frag<DH>(Vs, fo, dt, sub, 1), pf1, o[dt]);
        }
      }
    }
  }
  if (SUM_BY_MFMA) {       // O^T row d = DH sits in accumulator block DH / 32, register (DH % 32) / 8 * 4 of the hi = 0 lanes
    constexpr int blk = DH / 32, reg = (DH % 32) / 8 * 4;
    static_assert((DH % 32) % 8 == 0 && ((DH % 32) / 8) * 8 + 0 == DH % 32, "spare row must be row 8k of its block");
    l = o[blk][reg];
    l = __shfl(l, li, 64);                 // hi = 1 lanes take it from lane li
  }
  const float inv = 1.f / l;
  store_T_acc<DH>(o, inv, p.Out + b * p.bo + h * DH, p.ldo, q, p.T, hi);
  if (p.L && hi == 0 && q < p.T) p.L[((long long)b * p.H + h) * p.T + q] = m + log2f(l);
}

// ================================================================================================
// forward, long key ranges with a spare contraction slot (dh = 40), round 6: one wave = 64 queries, phased software pipeline
// ================================================================================================
// The kernel above runs, per wave and 32 x 32 score sub-tile, a strict chain  3 MFMA -> ~46 VALU (16 of them v_exp_f32) -> 4 MFMA:
// MFMA and VALU of ONE wave never overlap, and four waves per SIMD at random phases recover only part of it (the
// measured time is close to the SUM of the MFMA and the VALU issue time, 600 clk per sub-tile against 224 clk of matrix pipe).
// tools/probe/mfma_rate.hip: a SIMD hides ~18 clk (4 plain VALU) under every 32-clk MFMA when the VALU work is independent of it
// and sits next to it in program order.  This kernel arranges exactly that:
//  * a wave owns TWO 32-query blocks a, b.  Every K fragment (ds_read_b128) and every V^T fragment (ds_read_b64_tr_b16 pair)
//    read from LDS feeds two MFMAs, and a staged 64-key tile serves 256 queries instead of 128;
//  * the loop is a sequence of PHASES, each = the softmax VALU of one block on scores that are already complete, beside the
//    seven MFMAs of the OTHER block (its P.V of the previous sub-tile, then its scores of the next one), one MFMA in front of
//    every 4-instruction softmax chunk (pk_fma, 2 exp, cvt_pk), pinned with sched_barrier:
//        Y(j): softmax a(j)  ||  P.V b(j-1), then V^T(j) fragments, S b(j), then K(j+1) fragments
//        X(j): softmax b(j)  ||  P.V a(j),  S a(j+1)             (odd j: + the staged registers -> LDS, next global loads)
//  * a phase body has NO branch: no running-max pass in front of the exponentials.  P = exp2(s * scale2 - m) is formed with the
//    row's current m; the eight packed bf16 P words are OR-ed and bit 14 (exponent >= 128: p >= 2, inf, NaN) ballot-tested at the
//    end of the phase.  Only then (first tile of a row, a key that beats the running max by more than 2x) the slow path takes the
//    true row max, raises m, rescales O^T and recomputes the sub-tile's P — before any MFMA has consumed it (P.V of a sub-tile is
//    issued in the NEXT phase).  m, l and LSE = m + log2 l stay consistent, the result is the exact softmax;
//  * keys beyond S need no VALU mask: the K image's first pad column (d = DH) holds -29952 for such rows against 1.0 in the Q
//    fragment, so their scores come out of the MFMA as -29952 and p = 0; row sums ride in the spare O^T row (ones column of the
//    V image) as in the kernel above;
//  * two LDS tile buffers, ONE barrier per 64-key tile (in the odd Y phase, between the last read of the old tile and the first
//    read of the next one); tiles are fetched with the row offset in the bounds-checked VGPR offset, so rows beyond S are zeros.
#ifndef ATTN_FWD64
#define ATTN_FWD64 1
#endif
// attn_bwd_dkv64_kernel below is NOT the product path (0): measured on B16 H8 T = S = 4096 it takes ~1000 us against the ~880 us of
// attn_bwd_dkv_kernel<40, 3> (tools/ab_attn_bwd.py, profiles/r06_ab/attention_bwd64.txt).  It needs 365 registers, i.e. ONE wave per
// SIMD, and with one wave per SIMD nothing covers its LDS reads, tile staging and barrier skew (knock-out timings: MFMA stream alone
// 424 us, + VALU 570, staging 221, barrier 176): three resident waves of the old kernel hide exactly those.  Kept for -DATTN_BWD64=1 A/B.
#ifndef ATTN_BWD64
#define ATTN_BWD64 0
#endif
#ifndef ATTN_FWD64_OCC
#define ATTN_FWD64_OCC 2
#endif
#ifndef ATTN_FWD64_FENCE
#define ATTN_FWD64_FENCE 1
#endif
#ifndef ATTN_FWD64_ROWS
#define ATTN_FWD64_ROWS 128
#endif
#ifndef ATTN_FWD64_MIN_S
#define ATTN_FWD64_MIN_S 512
#endif
#define A64_FENCE() do { if (ATTN_FWD64_FENCE) __builtin_amdgcn_sched_barrier(0); } while (0)
// timing variants (results wrong in every variant but 0; tools/gpu_r06_b.sh): 1 exp2 -> plain add   2 no softmax VALU   3 no MFMA
//   4 no LDS fragment reads (8: V^T only, 9: K only)   5 no tile staging (LDS stores, global loads)   6 no tile barrier   7 = 2 + 4 + 5 + 6 (MFMA stream only)
#ifndef ATTN64_PROBE
#define ATTN64_PROBE 0
#endif
__device__ __forceinline__ f32x16 a64_mfma(bf16x8 a, bf16x8 b, f32x16 c) {
  if (ATTN64_PROBE == 3) { Frag fa, fb; fa.v = a; fb.v = b; c[0] += __builtin_bit_cast(float, fa.w[0] ^ fb.w[1]); return c; }
  return __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c, 0, 0, 0);
}
__device__ __forceinline__ float a64_exp2(float x) { return ATTN64_PROBE == 1 ? x + 1.f : fast_exp2(x); }

// TileRegs with the tile row offset in the per-lane (bounds-checked) offset: rows at or beyond the resource's extent read as zeros
template <int DH>
struct TileRegsV {
  using C = Cfg<DH>;
  uint4 v[C::NIT][2];
  unsigned vo[C::NIT], step[C::NIT];
  unsigned ldb;
  __device__ __forceinline__ void init(int ld) {
    ldb = ld * 2;
#pragma unroll
    for (int it = 0; it < C::NIT; ++it) {
      const int item = threadIdx.x + it * 256;
      const int ch = item >> 5, kp = item & 31;
      const bool on = item < 32 * C::NCH;
      vo[it] = on ? (unsigned)((2 * kp * ld + ch * 8) * 2) : 0xFFFF0000u;
      step[it] = on ? 64u * ldb : 0u;
    }
  }
  __device__ __forceinline__ void load(__amdgpu_buffer_rsrc_t rs) {      // the next 64-row tile
#pragma unroll
    for (int it = 0; it < C::NIT; ++it) {
      v[it][0] = buf_load16(rs, vo[it], 0);
      v[it][1] = buf_load16(rs, vo[it] + ldb, 0);
      vo[it] += step[it];
    }
  }
  // no divergent branch (a phase must stay ONE basic block): threads without an item store their (zero) registers into row
  // slots 8.. of the 256-byte image rows, which no fragment read of a head dim <= 56 touches
  __device__ __forceinline__ void store_rows(bf16_t* lds) const {
    static_assert(C::NCH + 1 <= 8 && C::LDE == 128 && 32 * C::NCH + 96 >= 256 * C::NIT, "idle threads need free slots 8..10");
#pragma unroll
    for (int it = 0; it < C::NIT; ++it) {
      const int item = threadIdx.x + it * 256;
      const int ch = item >> 5, kp = item & 31;
      const int slot = item < 32 * C::NCH ? ch : ch + (8 - C::NCH);
      *(uint4*)(lds + img_off<C::LDE>(2 * kp, slot)) = v[it][0];
      *(uint4*)(lds + img_off<C::LDE>(2 * kp + 1, slot)) = v[it][1];
    }
  }
};

// s * sc - mm on two adjacent score elements as ONE packed VALU instruction (the compiler scalarises the builtin form in most
// chunks of the phase below: +1 VALU and a hazard nop each)
#ifndef ATTN64_SCALAR_FMA
#define ATTN64_SCALAR_FMA (!ATTN_PK_DMA)
#endif
__device__ __forceinline__ f32x2 pk_fms(f32x2 s, f32x2 sc, f32x2 mm) {
  f32x2 d;
#if ATTN64_SCALAR_FMA
  asm("v_fma_f32 %0, %1, %2, -%3" : "=v"(d.x) : "v"(s.x), "s"(sc.x), "v"(mm.x));
  asm("v_fma_f32 %0, %1, %2, -%3" : "=v"(d.y) : "v"(s.y), "s"(sc.x), "v"(mm.x));
#else
  asm("v_pk_fma_f32 %0, %1, %2, %3 neg_lo:[0,0,1] neg_hi:[0,0,1]" : "=v"(d) : "v"(s), "s"(sc), "v"(mm));
#endif
  return d;
}

// ---- compact LDS image for head dims <= 56 (round 6): [64 rows][8 slots of 16 B] = 128-byte rows, filled by LDS-DMA ----
// slot L of row r lives at physical slot ((L>>2) ^ ((r>>1)&1)) << 2 | ((L&3) ^ ((r>>2)&3)) (an involution in L).  ds_read_b128 row
// fragments (16 lanes of distinct r mod 16, one L): bank slot 8 (r&1) + 4 (q ^ (r>>1&1)) + (j ^ (r>>2&3)) takes 16 distinct values;
// the transposing 4-row x 64-byte gathers (rows r0..r0+3, r0 % 4 == 0, one aligned block of 4 slots) hit the 4 bank quarters
// 2 (r&1) + (q ^ (r>>1&1)) once each.  One buffer_load ... lds instruction writes 1 KB = 8 image rows linearly (lane l -> row l >> 3,
// physical slot l & 7), so the swizzle is applied on the SOURCE side: each lane fetches the logical slot its physical slot holds;
// lanes of the pad / unused slots (L >= DH / 8) fetch out of range = zeros.
__device__ __forceinline__ int img64_slot(int r, int L) { return ((((L >> 2) ^ ((r >> 1) & 1)) << 2) | ((L & 3) ^ ((r >> 2) & 3))); }
__device__ __forceinline__ int img64_off(int r, int L) { return r * 64 + img64_slot(r, L) * 8; }      // elements
template <int DH>
struct FragOff64 {
  using C = Cfg<DH>;
  static_assert(C::DV <= 64, "128-byte image rows hold 64 columns");
  int row[C::NKS], tr_lo[C::NDT], tr_hi[C::NDT];
  __device__ __forceinline__ FragOff64() {
    const int lane = threadIdx.x & 63, li = lane & 31, hi = lane >> 5, il = lane & 15, d16 = (lane >> 4) & 1;
#pragma unroll
    for (int ks = 0; ks < C::NKS; ++ks) row[ks] = img64_off(li, ks * 2 + hi);
#pragma unroll
    for (int dt = 0; dt < C::NDT; ++dt) {
      const int L = dt * 4 + 2 * d16 + ((il & 3) >> 1), r = 4 * hi + (il >> 2);
      tr_lo[dt] = img64_off(r, L) + 4 * (il & 1);
      tr_hi[dt] = img64_off(r + 8, L) + 4 * (il & 1);
    }
  }
};
// The DMA is issued from inline asm, not through __builtin_amdgcn_raw_ptr_buffer_load_lds: hipcc (ROCm 7.2) puts its own
// s_waitcnt vmcnt(0) in front of the first ds_read that follows a builtin LDS-DMA (it cannot prove that the read does not alias the
// DMA's target), which turns a three-tile lead into none (measured: the 4-buffer ring was no faster than the 2-buffer one).  With
// the asm form the compiler sees no LDS-DMA; the counted attn_wait_vmcnt<N>() in front of each tile barrier is the only wait.
// Consequence: no compiler-counted VMEM load may live in a loop that uses these (its vmcnt would not include the asm pieces and
// would over-wait).  M0 = LDS byte address of the wave's 1-KB destination, written in the same statement that uses it.
typedef int i32x4_native __attribute__((ext_vector_type(4)));
struct DmaRsrc { i32x4_native w; };
__device__ __forceinline__ DmaRsrc make_dma_rsrc(const void* base, unsigned bytes) {
  const unsigned long long a = (unsigned long long)base;
  DmaRsrc r;
  r.w = i32x4_native{__builtin_amdgcn_readfirstlane((int)(unsigned)a), __builtin_amdgcn_readfirstlane((int)(unsigned)(a >> 32)),
                     __builtin_amdgcn_readfirstlane((int)bytes), 0x00020000};
  return r;
}
__device__ __forceinline__ unsigned lds_addr(const bf16_t* p) {
  return (unsigned)(unsigned long long)(__attribute__((address_space(3))) const void*)p;
}
__device__ __forceinline__ void dma16(const DmaRsrc& rs, unsigned voff, unsigned lds_wave_base_bytes) {
  asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tbuffer_load_dwordx4 %1, %2, 0 offen lds" ::"s"(lds_wave_base_bytes), "v"(voff), "s"(rs.w) : "memory");
}
__device__ __forceinline__ void dma4(const DmaRsrc& rs, unsigned voff, unsigned lds_wave_base_bytes) {      // 4 bytes per lane: 256 B per piece
  asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tbuffer_load_dword %1, %2, 0 offen lds" ::"s"(lds_wave_base_bytes), "v"(voff), "s"(rs.w) : "memory");
}
template <int N>
__device__ __forceinline__ void attn_wait_vmcnt() { asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory"); }
// this wave's share (two of the eight 1-KB pieces) of a 64-row tile image; the row offset lives in the bounds-checked lane offset
template <int DH, int ROWS = 64>
struct TileDma {
  static constexpr int NP = ROWS / 32;             // 1-KB pieces (8 image rows each) per wave of a 4-wave workgroup
  unsigned vo[NP], step[NP];
  __device__ __forceinline__ void init(int ld) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
#pragma unroll
    for (int i = 0; i < NP; ++i) {
      const int r = 8 * (NP * wave + i) + (lane >> 3), L = img64_slot(r, lane & 7);
      const bool on = L < Cfg<DH>::NCH;
      vo[i] = on ? (unsigned)((r * ld + L * 8) * 2) : 0x80000000u;
      step[i] = on ? (unsigned)(ROWS * ld * 2) : 0u;
    }
  }
  __device__ __forceinline__ void issue(const DmaRsrc& rs, bf16_t* img) {      // the next tile -> img
    const unsigned base = __builtin_amdgcn_readfirstlane(lds_addr(img) + (threadIdx.x >> 6) * (NP * 1024));
#pragma unroll
    for (int i = 0; i < NP; ++i) {
      dma16(rs, vo[i], base + i * 1024);
      vo[i] += step[i];
    }
  }
};

// key-validity column of a K image: pad slot (d = DH .. DH + 7) of row r = {valid ? 0 : -29952, 0, ...}; four threads per row
// write the same value (no divergent branch inside a phase)
template <int DH>
__device__ __forceinline__ void write_key_mask(bf16_t* kimg, int row0, int S) {
  const int r = threadIdx.x & 63;
  *(uint4*)(kimg + img_off<Cfg<DH>::LDE>(r, Cfg<DH>::NCH)) = make_uint4(row0 + r < S ? 0u : 0xC6EAu, 0u, 0u, 0u);
}

// Pad slots of a freshly landed tile: key-validity column of the K image, ones column of the V image (the DMA zero-fills them).
// EVERY WAVE WRITES ONLY THE ROWS ITS OWN DMA PIECES COVER (rows [ROWS / 4 * wave, + ROWS / 4) of both images), behind its own vmcnt
// wait: a pad written into a row whose piece is still in flight from ANOTHER wave is overwritten with zeros when that piece lands
// (round 6: found as a one-in-hundreds-of-workgroups wrong dK / dV at three workgroups per CU; the same write pattern had been
// latent here, hidden by the long DMA lead).
template <int DH, int ROWS>
__device__ __forceinline__ void fwd64_write_pads(bf16_t* tile, int row0, int S) {
  constexpr int IMG = ROWS * 64, PER = ROWS / 4;               // rows per wave
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int img = lane / PER, r = PER * wave + lane % PER;      // lanes [0, PER): K image, [PER, 2 PER): V image
  if (img < 2) {
    const unsigned w0 = img ? 0x3F80u : (row0 + r < S ? 0u : 0xC6EAu);
    *(uint4*)(tile + img * IMG + img64_off(r, Cfg<DH>::NCH)) = make_uint4(w0, 0u, 0u, 0u);
  }
}

// One phase (see above).  x = the block whose scores sx become P (px); y = the other block: its pending P.V and its next scores.
//   LOADS: Y phase — reload the V^T fragments from lds + VOFF after the P.V MFMAs and the K fragments from lds + KOFF after the
//          score MFMAs.  PADS >= 0 (odd Y phase): in front of the K reload, the tile barrier — this wave's DMA pieces of the NEXT tile
//          (issued THREE tiles earlier into the images at lds + PADS / + PADS + IMG) have landed (counted vmcnt: the 8 pieces of the two
//          tiles behind it stay in flight), its pad slots are written (key validity column of K, ones column of V: the DMA zero-fills
//          them), then s_barrier.
//   STAGE >= 0: X phase that issues the DMA of the tile FOUR ahead into this tile's own images at lds + STAGE / + STAGE + IMG (nobody
//          reads them any more: the barrier of the preceding Y phase).  Four tile buffers: a tile period is ~1 us, about the latency of
//          an LDS-DMA piece under load — with two buffers (one tile of lead) the wait in front of the barrier cost ~90 us per launch.
template <int DH, int ROWS, bool LOADS, int VOFF, int KOFF, int PADS, int STAGE>
__device__ __forceinline__ void fwd64_phase(const f32x2 sc, const FragOff64<DH>& fo, bf16_t* lds,
                                            const f32x16& sx, f32x2& mm, Frag (&px)[2], f32x16 (&ox)[Cfg<DH>::NDT],
                                            f32x16& sy, const Frag (&py)[2], f32x16 (&oy)[Cfg<DH>::NDT], const bf16x8 (&qfy)[Cfg<DH>::NKS],
                                            bf16x8 (&vfr)[Cfg<DH>::NDT][2], bf16x8 (&kfr)[Cfg<DH>::NKS],
                                            TileDma<DH, ROWS>& kd, TileDma<DH, ROWS>& vd, const DmaRsrc& rsK, const DmaRsrc& rsV, const int pad_row0,
                                            const int S) {
  using C = Cfg<DH>;
  constexpr int NPV = 2 * C::NDT, NM = NPV + C::NKS, IMG = ROWS * 64;
  f32x2 t = pk_fms(f32x2{sx[0], sx[1]}, sc, mm);   // the exp argument of chunk c is formed in chunk c - 1 (no dependent back-to-back VALU)
  float pe0 = 0.f, pe1 = 0.f;
#pragma unroll
  for (int c = 0; c < 8; ++c) {
#pragma unroll
    for (int i = (c * NM + 7) / 8; i < ((c + 1) * NM + 7) / 8; ++i) {
      if (i < NPV) {                               // k-step outer: neighbouring MFMAs accumulate into different O^T blocks
        const int k2 = i / C::NDT, dt = i % C::NDT;
        oy[dt] = a64_mfma(vfr[dt][k2], py[k2].v, oy[dt]);
      } else {
        const int ks = i - NPV;
        f32x16 z;
#pragma unroll
        for (int r = 0; r < 16; ++r) z[r] = 0.f;
        sy = a64_mfma(kfr[ks], qfy[ks], ks == 0 ? z : sy);
      }
      A64_FENCE();
      if (LOADS && i == NPV - 1 && ATTN64_PROBE != 4 && ATTN64_PROBE != 7 && ATTN64_PROBE != 8) {      // every V^T fragment has been consumed: fetch the next sub-tile's
#pragma unroll
        for (int k2 = 0; k2 < 2; ++k2)
#pragma unroll
          for (int dt = 0; dt < C::NDT; ++dt) {
            const bf16_t* vb = lds + VOFF + k2 * 16 * 64;
            vfr[dt][k2] = tr_frag(vb + fo.tr_lo[dt], vb + fo.tr_hi[dt]);
          }
        A64_FENCE();
      }
      if (LOADS && i == NM - 1 && PADS >= 0 && ATTN64_PROBE != 6 && ATTN64_PROBE != 7) {
        if (ROWS == 64) attn_wait_vmcnt<8>();      // this wave's 4 pieces of the next tile have landed; the two tiles behind it stay in flight
        else attn_wait_vmcnt<0>();                  // 128-row tiles, two buffers: the next tile (issued one tile ago) is the only one in flight
        fwd64_write_pads<DH, ROWS>(lds + PADS, pad_row0, S);
        __syncthreads();                           // the next tile's images are complete; nobody reads the old tile any more
      }
      if (LOADS && i == NM - 1 && (ATTN64_PROBE == 4 || ATTN64_PROBE == 7 || ATTN64_PROBE == 9)) {      // (opaque: the score MFMAs must not become loop-invariant)
#pragma unroll
        for (int ks = 0; ks < C::NKS; ++ks) asm volatile("" : "+v"(kfr[ks]));
      }
      if (LOADS && i == NM - 1 && ATTN64_PROBE != 4 && ATTN64_PROBE != 7 && ATTN64_PROBE != 9) {
#pragma unroll
        for (int ks = 0; ks < C::NKS; ++ks) kfr[ks] = *(const bf16x8*)(lds + KOFF + fo.row[ks]);
        A64_FENCE();
      }
    }
    if (ATTN64_PROBE == 2 || ATTN64_PROBE == 7) {
      px[c >> 2].w[c & 3] = __builtin_bit_cast(uint32_t, sx[2 * c]);
    } else {                                       // softmax chunk c: score elements 2c, 2c + 1 of this lane's query
      const float e0 = a64_exp2(t.x), e1 = a64_exp2(t.y);
      if (c < 7) t = pk_fms(f32x2{sx[2 * c + 2], sx[2 * c + 3]}, sc, mm);
      if (c > 0) px[(c - 1) >> 2].w[(c - 1) & 3] = pack2bf(pe0, pe1);      // the previous chunk's pair: no trans -> VALU hazard nop
      pe0 = e0; pe1 = e1;
    }
    A64_FENCE();
    if (STAGE >= 0 && ATTN64_PROBE != 5 && ATTN64_PROBE != 7) {
      if (c == 0) { kd.issue(rsK, lds + STAGE); A64_FENCE(); }
      if (c == 1) { vd.issue(rsV, lds + STAGE + IMG); A64_FENCE(); }
    }
  }
  if (ATTN64_PROBE != 2 && ATTN64_PROBE != 7) px[1].w[3] = pack2bf(pe0, pe1);
  const uint32_t orv = (px[0].w[0] | px[0].w[1] | px[0].w[2]) | (px[0].w[3] | px[1].w[0] | px[1].w[1]) | (px[1].w[2] | px[1].w[3]);
  if (ATTN64_PROBE == 0 && __builtin_amdgcn_ballot_w64((orv & 0x40004000u) != 0u) != 0) {      // some p >= 2 (or inf / NaN): raise the running max
    float mv = sx[0];
#pragma unroll
    for (int r = 1; r < 16; ++r) mv = fmaxf(mv, sx[r]);
    mv = xhalf_max(mv) * sc.x;                     // scale2 > 0
    const float mn = fmaxf(mm.x, mv);
    const float alpha = fast_exp2(mm.x - mn);
    mm = f32x2{mn, mn};
#pragma unroll
    for (int dt = 0; dt < C::NDT; ++dt)
#pragma unroll
      for (int r = 0; r < 16; ++r) ox[dt][r] *= alpha;
#pragma unroll
    for (int c = 0; c < 8; ++c) {
      const f32x2 u = pk_fma<ATTN_PK_DMA>(f32x2{sx[2 * c], sx[2 * c + 1]}, sc, -mm);
      px[c >> 2].w[c & 3] = pack2bf(fast_exp2(u.x), fast_exp2(u.y));
    }
  }
}

template <int DH, int ROWS>
__global__ __launch_bounds__(256, ATTN_FWD64_OCC) void attn_fwd64_kernel(AttnArgs p) {
  using C = Cfg<DH>;
  static_assert(DH % 16 == 8 && C::DV > DH && C::DV <= 64, "needs a spare contraction slot (key mask), a spare O^T row (row sum), 128-byte image rows");
  static_assert(ROWS == 64 || ROWS == 128, "64-key tiles in a ring of four, or 128-key tiles in two buffers");
  constexpr int IMG = ROWS * 64, TILE = 2 * IMG, NBUF = 256 / ROWS;    // one image = ROWS rows x 128 bytes; a tile = K image + V image
  __shared__ __attribute__((aligned(1024))) bf16_t lds[NBUF * TILE];   // 64 KB either way
  const FragOff64<DH> fo;
  const Blk blk = xcd_block(p.xcd_raster);
  const int b = blk.b, h = blk.h;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int li = lane & 31, hi = lane >> 5;
  const int q0 = blk.x * 256 + wave * 64 + li;                         // block a: row q0, block b: row q0 + 32
  const bf16_t* Qb = p.Q + b * p.bq + h * DH;
  const bf16_t* Kb = p.K + b * p.bk + h * DH;
  const bf16_t* Vb = p.V + b * p.bv + h * DH;
  const f32x2 sc = {p.scale2, p.scale2};

  TileDma<DH, ROWS> kd, vd;
  const DmaRsrc rsK = make_dma_rsrc(Kb, (unsigned)(((long long)(p.S - 1) * p.ldk + DH) * 2));
  const DmaRsrc rsV = make_dma_rsrc(Vb, (unsigned)(((long long)(p.S - 1) * p.ldv + DH) * 2));
  kd.init(p.ldk); vd.init(p.ldv);
#pragma unroll
  for (int tb = 0; tb < NBUF; ++tb) { kd.issue(rsK, lds + tb * TILE); vd.issue(rsV, lds + tb * TILE + IMG); }      // the first NBUF tiles

  bf16x8 qf[2][C::NKS];
  load_row_frags<DH>(Qb, p.ldq, q0, p.T, hi, qf[0]);
  load_row_frags<DH>(Qb, p.ldq, q0 + 32, p.T, hi, qf[1]);
  if (hi == 1) {                                   // d = DH (first pad slot of k-step DH / 16) = 1.0: picks up the key-validity column
    Frag t;
    t.q = make_uint4(0x3F80u, 0u, 0u, 0u);
    qf[0][DH / 16] = t.v; qf[1][DH / 16] = t.v;
  }

  f32x16 oa[C::NDT], ob[C::NDT], sa, sb;
  Frag pa[2], pb[2];
  bf16x8 vfr[C::NDT][2], kfr[C::NKS];
#pragma unroll
  for (int dt = 0; dt < C::NDT; ++dt)
#pragma unroll
    for (int r = 0; r < 16; ++r) { oa[dt][r] = 0.f; ob[dt][r] = 0.f; }
  {
    Frag z;
    z.q = make_uint4(0u, 0u, 0u, 0u);
    pa[0] = z; pa[1] = z; pb[0] = z; pb[1] = z;
#pragma unroll
    for (int dt = 0; dt < C::NDT; ++dt) { vfr[dt][0] = z.v; vfr[dt][1] = z.v; }
  }
  f32x2 ma = {-1e30f, -1e30f}, mb = ma;      // running row max (log2 units), duplicated for the packed fma
  attn_wait_vmcnt<0>();                           // (one-off; also covers the Q fragments)
  fwd64_write_pads<DH, ROWS>(lds, 0, p.S);        // tile 0's pad slots (every later tile gets its own at its publishing barrier)
  __syncthreads();
#pragma unroll
  for (int ks = 0; ks < C::NKS; ++ks) kfr[ks] = *(const bf16x8*)(lds + fo.row[ks]);
#pragma unroll
  for (int r = 0; r < 16; ++r) sa[r] = 0.f;
#pragma unroll
  for (int ks = 0; ks < C::NKS; ++ks) sa = __builtin_amdgcn_mfma_f32_32x32x16_bf16(kfr[ks], qf[0][ks], sa, 0, 0, 0);

  constexpr int NSUB = ROWS / 32;
  const int ntiles = (p.S + ROWS - 1) / ROWS;
  // one ROWS-key tile in buffer TB: sub-tiles NSUB t .. NSUB t + NSUB - 1 (a ragged last tile is processed whole: its absent keys are
  // masked).  The last sub-tile's Y phase carries the tile barrier, its X phase the DMA of the tile NBUF ahead into this tile's images.
  auto tile = [&](auto tb_, int t) __attribute__((always_inline)) {
    constexpr int TB = decltype(tb_)::value;
    constexpr int KI = TB * TILE, VI = KI + IMG, KN = ((TB + 1) % NBUF) * TILE;
#define F64_Y sa, ma, pa, oa, sb, pb, ob, qf[1], vfr, kfr, kd, vd, rsK, rsV
#define F64_X sb, mb, pb, ob, sa, pa, oa, qf[0], vfr, kfr, kd, vd, rsK, rsV
    fwd64_phase<DH, ROWS, true, VI, KI + 32 * 64, -1, -1>(sc, fo, lds, F64_Y, 0, p.S);
    fwd64_phase<DH, ROWS, false, 0, 0, -1, -1>(sc, fo, lds, F64_X, 0, p.S);
    if constexpr (NSUB == 4) {
      fwd64_phase<DH, ROWS, true, VI + 32 * 64, KI + 64 * 64, -1, -1>(sc, fo, lds, F64_Y, 0, p.S);
      fwd64_phase<DH, ROWS, false, 0, 0, -1, -1>(sc, fo, lds, F64_X, 0, p.S);
      fwd64_phase<DH, ROWS, true, VI + 64 * 64, KI + 96 * 64, -1, -1>(sc, fo, lds, F64_Y, 0, p.S);
      fwd64_phase<DH, ROWS, false, 0, 0, -1, -1>(sc, fo, lds, F64_X, 0, p.S);
    }
    fwd64_phase<DH, ROWS, true, VI + (ROWS - 32) * 64, KN, KN, -1>(sc, fo, lds, F64_Y, (t + 1) * ROWS, p.S);
    fwd64_phase<DH, ROWS, false, 0, 0, -1, KI>(sc, fo, lds, F64_X, 0, p.S);
#undef F64_Y
#undef F64_X
  };
  for (int t = 0; t < ntiles; t += NBUF) {
    tile(std::integral_constant<int, 0>{}, t);
    if (t + 1 < ntiles) tile(std::integral_constant<int, 1>{}, t + 1);
    if constexpr (NBUF == 4) {
      if (t + 2 < ntiles) tile(std::integral_constant<int, 2>{}, t + 2);
      if (t + 3 < ntiles) tile(std::integral_constant<int, 3>{}, t + 3);
    }
  }
  attn_wait_vmcnt<0>();                           // (DMA pieces of tiles beyond the end — zeros — must not outlive the workgroup's LDS)
#pragma unroll
  for (int k2 = 0; k2 < 2; ++k2)                   // the last sub-tile's P.V of block b
#pragma unroll
    for (int dt = 0; dt < C::NDT; ++dt) ob[dt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(vfr[dt][k2], pb[k2].v, ob[dt], 0, 0, 0);

  constexpr int sblk = DH / 32, sreg = (DH % 32) / 8 * 4;              // O^T row d = DH: the row sum (hi = 0 lanes)
  float la = __shfl(oa[sblk][sreg], li, 64), lb = __shfl(ob[sblk][sreg], li, 64);
  bf16_t* Ob = p.Out + b * p.bo + h * DH;
  int qrow = q0;
  asm volatile("" : "+v"(qrow));
  store_T_acc<DH>(oa, 1.f / la, Ob, p.ldo, qrow, p.T, hi);
  store_T_acc<DH>(ob, 1.f / lb, Ob, p.ldo, qrow + 32, p.T, hi);
  if (p.L && hi == 0) {
    float* Lb = p.L + ((long long)b * p.H + h) * p.T;
    if (qrow < p.T) Lb[qrow] = ma.x + log2f(la);
    if (qrow + 32 < p.T) Lb[qrow + 32] = mb.x + log2f(lb);
  }
}

// ================================================================================================
// backward: Delta[b][h][q] = sum_d dO * O — computed in the PROLOGUE of the dQ kernel (round 5; was a kernel of its own: 13 launches
// per step that read O and dO once more).  The dQ kernel holds its 32 queries' dO rows as MFMA fragments anyway — lane (li, hi) owns
// the 8-element slices [16 ks + 8 hi, +8) of row q = li — so Delta is the same slices of O multiplied in, summed per lane and across
// the two halves of the wave.  It is written to p.Delta for the dK/dV kernel, which therefore runs AFTER the dQ kernel (launch_bwd).
// ================================================================================================
template <int DH>
__device__ __forceinline__ float row_delta(const bf16_t* o_base, int ldo, int row, int R, int hi, const bf16x8* dof) {
  bf16x8 of[Cfg<DH>::NKS];
  load_row_frags<DH>(o_base, ldo, row, R, hi, of);
  float s = 0.f;
#pragma unroll
  for (int ks = 0; ks < Cfg<DH>::NKS; ++ks) {
    Frag a, g;
    a.v = of[ks]; g.v = dof[ks];
    float x[8], y[8];
    unpack8(a.q, x);
    unpack8(g.q, y);
#pragma unroll
    for (int k = 0; k < 8; ++k) s = fmaf(x[k], y[k], s);
  }
  return s + __shfl_xor(s, 32, 64);
}

// ================================================================================================
// backward dQ: one wave = 32 queries, loop over key tiles
// ================================================================================================
template <int DH>
__global__ __launch_bounds__(256, (DH <= 64 ? 3 : DH <= 80 ? 2 : 1)) void attn_bwd_dq_kernel(AttnArgs p) {
  using C = Cfg<DH>;
  __shared__ __attribute__((aligned(16))) bf16_t Ks[64 * C::LDE];
  __shared__ __attribute__((aligned(16))) bf16_t Vs[64 * C::LDE];
  const FragOff<DH> fo;
  const Blk blk = xcd_block(p.xcd_raster);
  const int b = blk.b, h = blk.h;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int li = lane & 31, hi = lane >> 5;
  const int q = blk.x * 128 + wave * 32 + li;
  const bf16_t* Kb = p.K + b * p.bk + h * DH;
  const bf16_t* Vb = p.V + b * p.bv + h * DH;

  bf16x8 qf[C::NKS], dof[C::NKS];
  load_row_frags<DH>(p.Q + b * p.bq + h * DH, p.ldq, q, p.T, hi, qf);
  load_row_frags<DH>(p.dO + b * p.bo + h * DH, p.ldo, q, p.T, hi, dof);
  float Lq = 0.f;
  const float Dq = row_delta<DH>(p.O + b * p.bo + h * DH, p.ldo, q, p.T, hi, dof);      // (0 for q >= T: both fragments are zero there)
  if (q < p.T) {
    Lq = p.L[((long long)b * p.H + h) * p.T + q];
    if (hi == 0) {                                  // for the dK/dV kernel, launched behind this one
      p.Delta[((long long)b * p.H + h) * p.T + q] = Dq;
      if (p.LD) *(float2*)(p.LD + 2 * (((long long)b * p.H + h) * p.T + q)) = make_float2(Lq, Dq);
    }
  }

  f32x16 acc[C::NDT];
#pragma unroll
  for (int dt = 0; dt < C::NDT; ++dt)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[dt][r] = 0.f;

  TileRegs<DH> kr, vr;
  const __amdgpu_buffer_rsrc_t rsK = make_rsrc(Kb, (unsigned)(((long long)(p.S - 1) * p.ldk + DH) * 2));
  const __amdgpu_buffer_rsrc_t rsV = make_rsrc(Vb, (unsigned)(((long long)(p.S - 1) * p.ldv + DH) * 2));
  kr.init(p.ldk); vr.init(p.ldv);
  kr.load(rsK, 0);
  vr.load(rsV, 0);
  zero_pad_cols<DH>(Ks);
  zero_pad_cols<DH>(Vs);
  for (int kv0 = 0; kv0 < p.S; kv0 += 64) {
    __syncthreads();
    kr.store_rows(Ks);
    vr.store_rows(Vs);
    __syncthreads();
    if (kv0 + 64 < p.S) {
      kr.load(rsK, kv0 + 64);
      vr.load(rsV, kv0 + 64);
    }
    const int nsub = (p.S - kv0 > 32) ? 2 : 1;
    // one 32-row sub-tile; at dh <= 48 the two sub-tiles are separate instantiations (fragment addresses become immediates,
    // no per-read address VALU), larger head dims keep the runtime loop (the unrolled form would spill)
    auto sub_tile = [&](auto sub_) __attribute__((always_inline)) {
      const int sub = sub_;
      f32x16 s, dp;
#pragma unroll
      for (int r = 0; r < 16; ++r) { s[r] = 0.f; dp[r] = 0.f; }
#pragma unroll
      for (int ks = 0; ks < C::NKS; ++ks) {
        s = __builtin_amdgcn_mfma_f32_32x32x16_bf16(load_row_frag<DH>(Ks, fo, sub, ks), qf[ks], s, 0, 0, 0);
        dp = __builtin_amdgcn_mfma_f32_32x32x16_bf16(load_row_frag<DH>(Vs, fo, sub, ks), dof[ks], dp, 0, 0, 0);
      }
      float ds[16];
      if (kv0 + 64 > p.S || p.causal) {    // ragged last tile / causal mask
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const int key = kv0 + sub * 32 + acc_row(r, hi);
          const float pv = (key < p.S && (!p.causal || key <= q)) ? fast_exp2(s[r] * p.scale2 - Lq) : 0.f;
          ds[r] = pv * (dp[r] - Dq);
        }
      } else {
#pragma unroll
        for (int r = 0; r < 16; r += 2) {
          if (ATTN_PACKED && DH <= 48) {      // (dh 64 would spill)
            const f32x2 t = pk_fma<ATTN_PK_OLD>(f32x2{s[r], s[r + 1]}, f32x2{p.scale2, p.scale2}, f32x2{-Lq, -Lq});
            const f32x2 d = pk_fma<ATTN_PK_OLD>(f32x2{Dq, Dq}, f32x2{-1.f, -1.f}, f32x2{dp[r], dp[r + 1]});      // dp - Dq, kept packed
            const f32x2 o2 = pk_mul<ATTN_PK_OLD>(f32x2{fast_exp2(t.x), fast_exp2(t.y)}, d);
            ds[r] = o2.x; ds[r + 1] = o2.y;
          } else {
            ds[r] = fast_exp2(s[r] * p.scale2 - Lq) * (dp[r] - Dq);
            ds[r + 1] = fast_exp2(s[r + 1] * p.scale2 - Lq) * (dp[r + 1] - Dq);
          }
        }
      }
      const bf16x8 f0 = pack_acc(ds, 0), f1 = pack_acc(ds, 1);
#pragma unroll
      for (int dt = 0; dt < C::NDT; ++dt) {
        acc[dt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(load_T_frag<DH>(Ks, fo, dt, sub, 0), f0, acc[dt], 0, 0, 0);
        acc[dt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(load_T_frag<DH>(Ks, fo, dt, sub, 1), f1, acc[dt], 0, 0, 0);
      }
    };
    if constexpr (DH <= 48) {
      sub_tile(std::integral_constant<int, 0>{});
      if (nsub > 1) sub_tile(std::integral_constant<int, 1>{});
    } else {
#pragma unroll 1
      for (int sub = 0; sub < nsub; ++sub) sub_tile(sub);
    }
  }
  int qrow = q;
  asm volatile("" : "+v"(qrow));      // output addresses computed here, not carried through the loop (see the dK/dV kernel)
  store_T_acc<DH>(acc, p.scale, p.dQ + b * p.bq + h * DH, p.ldq, qrow, p.T, hi);
}

// ================================================================================================
// backward dQ with LDS-DMA tile staging (round 6, dh = 40, non-causal): attn_bwd_dq_kernel's arithmetic and occupancy, the K / V
// tiles arriving by buffer_load ... lds into a ring of three compact tile buffers two tiles ahead (see attn_bwd_dkv_dma_kernel).
// The DMA zero-fills the pad columns and the rows beyond S; nothing but the DMA pieces is a VMEM operation inside the loop.
// (Needs the TileDma / FragOff64 helpers defined with the forward kernel above.)
// ================================================================================================
#ifndef ATTN_DQ_DMA
#define ATTN_DQ_DMA 1
#endif

template <int DH>
__global__ __launch_bounds__(256, 3) void attn_bwd_dq_dma_kernel(AttnArgs p) {
  using C = Cfg<DH>;
  static_assert(C::DV <= 64, "compact images hold 64 columns");
  constexpr int IMG = 64 * 64, TILE = 2 * IMG, NB = 3;
  __shared__ __attribute__((aligned(1024))) bf16_t lds[NB * TILE];    // ring of [K image][V image]: 48 KB
  const FragOff64<DH> fo;
  const Blk blk = xcd_block(p.xcd_raster);
  const int b = blk.b, h = blk.h;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int li = lane & 31, hi = lane >> 5;
  const int q = blk.x * 128 + wave * 32 + li;
  const bf16_t* Kb = p.K + b * p.bk + h * DH;
  const bf16_t* Vb = p.V + b * p.bv + h * DH;

  TileDma<DH> kd, vd;
  const DmaRsrc rsK = make_dma_rsrc(Kb, (unsigned)(((long long)(p.S - 1) * p.ldk + DH) * 2));
  const DmaRsrc rsV = make_dma_rsrc(Vb, (unsigned)(((long long)(p.S - 1) * p.ldv + DH) * 2));
  kd.init(p.ldk); vd.init(p.ldv);
  kd.issue(rsK, lds); vd.issue(rsV, lds + IMG);                        // tile 0
  kd.issue(rsK, lds + TILE); vd.issue(rsV, lds + TILE + IMG);          // tile 1

  bf16x8 qf[C::NKS], dof[C::NKS];
  load_row_frags<DH>(p.Q + b * p.bq + h * DH, p.ldq, q, p.T, hi, qf);
  load_row_frags<DH>(p.dO + b * p.bo + h * DH, p.ldo, q, p.T, hi, dof);
  float Lq = 0.f;
  const float Dq = row_delta<DH>(p.O + b * p.bo + h * DH, p.ldo, q, p.T, hi, dof);
  if (q < p.T) {
    Lq = p.L[((long long)b * p.H + h) * p.T + q];
    if (hi == 0) {
      p.Delta[((long long)b * p.H + h) * p.T + q] = Dq;
      if (p.LD) *(float2*)(p.LD + 2 * (((long long)b * p.H + h) * p.T + q)) = make_float2(Lq, Dq);
    }
  }
  f32x16 acc[C::NDT];
#pragma unroll
  for (int dt = 0; dt < C::NDT; ++dt)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[dt][r] = 0.f;
  // (the prologue's compiler-counted loads / stores above were issued BEHIND the pieces of tiles 0 and 1 and share the in-order
  // counter with them: the first counted wait below is merely stricter than needed)
  const int ntiles = (p.S + 63) >> 6;

  auto tile = [&](auto tb_, int t) __attribute__((always_inline)) {
    constexpr int TB = decltype(tb_)::value;
    constexpr int KI = TB * TILE, VI = KI + IMG, NXT = ((TB + 2) % NB) * TILE;      // tile t + 2 goes where tile t - 1 was
    attn_wait_vmcnt<4>();                          // tile t landed; tile t + 1's 4 pieces stay in flight
    __syncthreads();
    kd.issue(rsK, lds + NXT); vd.issue(rsV, lds + NXT + IMG);
    const int kv0 = t * 64;
    auto sub_tile = [&](auto sub_) __attribute__((always_inline)) {
      constexpr int sub = decltype(sub_)::value, SO = sub * 32 * 64;
      f32x16 s, dp;
#pragma unroll
      for (int r = 0; r < 16; ++r) { s[r] = 0.f; dp[r] = 0.f; }
#pragma unroll
      for (int ks = 0; ks < C::NKS; ++ks) {
        s = __builtin_amdgcn_mfma_f32_32x32x16_bf16(*(const bf16x8*)(lds + KI + SO + fo.row[ks]), qf[ks], s, 0, 0, 0);
        dp = __builtin_amdgcn_mfma_f32_32x32x16_bf16(*(const bf16x8*)(lds + VI + SO + fo.row[ks]), dof[ks], dp, 0, 0, 0);
      }
      float ds[16];
      if (kv0 + 64 > p.S) {                // ragged last tile
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const int key = kv0 + sub * 32 + acc_row(r, hi);
          const float pv = key < p.S ? fast_exp2(s[r] * p.scale2 - Lq) : 0.f;
          ds[r] = pv * (dp[r] - Dq);
        }
      } else {
#pragma unroll
        for (int r = 0; r < 16; r += 2) {
          const f32x2 t2 = pk_fma<ATTN_PK_DMA>(f32x2{s[r], s[r + 1]}, f32x2{p.scale2, p.scale2}, f32x2{-Lq, -Lq});
          const f32x2 d = pk_fma<ATTN_PK_DMA>(f32x2{Dq, Dq}, f32x2{-1.f, -1.f}, f32x2{dp[r], dp[r + 1]});      // dp - Dq, kept packed
          const f32x2 o2 = pk_mul<ATTN_PK_DMA>(f32x2{fast_exp2(t2.x), fast_exp2(t2.y)}, d);
          ds[r] = o2.x; ds[r + 1] = o2.y;
        }
      }
      const bf16x8 f0 = pack_acc(ds, 0), f1 = pack_acc(ds, 1);
#pragma unroll
      for (int dt = 0; dt < C::NDT; ++dt) {
        const bf16_t* k0 = lds + KI + SO, *k1 = k0 + 16 * 64;
        acc[dt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(tr_frag(k0 + fo.tr_lo[dt], k0 + fo.tr_hi[dt]), f0, acc[dt], 0, 0, 0);
        acc[dt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(tr_frag(k1 + fo.tr_lo[dt], k1 + fo.tr_hi[dt]), f1, acc[dt], 0, 0, 0);
      }
    };
    sub_tile(std::integral_constant<int, 0>{});
    if (p.S - kv0 > 32) sub_tile(std::integral_constant<int, 1>{});
  };
  for (int t = 0; t < ntiles; t += 3) {
    tile(std::integral_constant<int, 0>{}, t);
    if (t + 1 < ntiles) tile(std::integral_constant<int, 1>{}, t + 1);
    if (t + 2 < ntiles) tile(std::integral_constant<int, 2>{}, t + 2);
  }
  attn_wait_vmcnt<0>();                           // (DMA pieces of tiles beyond the end must not outlive the workgroup's LDS)
  int qrow = q;
  asm volatile("" : "+v"(qrow));
  store_T_acc<DH>(acc, p.scale, p.dQ + b * p.bq + h * DH, p.ldq, qrow, p.T, hi);
}

// ================================================================================================
// backward dK, dV: one wave = 32 keys, loop over query tiles
// ================================================================================================
// OCC = workgroups per CU the register allocation is bounded for.  OCC = 3 (dh <= 64 only) gives up the register prefetch of the
// next Q / dO tile to fit 168 VGPRs: three waves per SIMD instead of two cover the tile loads' latency by occupancy instead.
template <int DH, int OCC>
__global__ __launch_bounds__(256, OCC) void attn_bwd_dkv_kernel(AttnArgs p) {
  using C = Cfg<DH>;
  __shared__ __attribute__((aligned(16))) bf16_t Qs[64 * C::LDE];
  __shared__ __attribute__((aligned(16))) bf16_t dOs[64 * C::LDE];
  const FragOff<DH> fo;
  __shared__ __attribute__((aligned(16))) float Ls[64], Dls[64];
  const Blk blk = xcd_block(p.xcd_raster);
  const int b = blk.b, h = blk.h;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int li = lane & 31, hi = lane >> 5;
  // blk.x = split * (key blocks) + key block; one split (the default) covers all queries
  const int nkb = (p.S + 127) >> 7;
  const int ts = p.tsplit > 1 ? blk.x / nkb : 0;
  const int key = (blk.x - ts * nkb) * 128 + wave * 32 + li;
  const int t_begin = ts * p.tchunk, t_end = p.tsplit > 1 ? min(p.T, t_begin + p.tchunk) : p.T;
  const bf16_t* Qb = p.Q + b * p.bq + h * DH;
  const bf16_t* dOb = p.dO + b * p.bo + h * DH;

  bf16x8 kf[C::NKS], vf[C::NKS];
  load_row_frags<DH>(p.K + b * p.bk + h * DH, p.ldk, key, p.S, hi, kf);
  load_row_frags<DH>(p.V + b * p.bv + h * DH, p.ldv, key, p.S, hi, vf);
  // FOLD (head dims with >= 3 spare contraction slots: 40 -> 48): the per-query terms ride in the padding of the two score
  // GEMMs.  The Q image carries -L/scale2 split into three bf16 pieces (24 bits) in columns DH..DH+2 and the K fragment holds
  // ones there, so S'^T = K.Q^T - L/scale2 comes out of the MFMA and p = exp2(scale2 * S'); likewise dO carries -Delta against
  // ones in the V fragment, so the MFMA yields dP - Delta.  That removes 8 ds_read_b128 of L / Delta (1 KB of VGPR return each,
  // broadcast or not) and 16 subtractions per 32 x 32 sub-tile from a kernel bound by VALU issue + LDS return bandwidth
  // (tools/probe/mfma_rate.hip, attn_probe.hip).  The transposed reads of those image columns land in accumulator rows
  // d = DH..DH+2 of dK^T / dV^T, which are never stored.
  constexpr bool FOLD = ATTN_DKV_FOLD && C::DK - DH >= 3 && DH % 8 == 0;
  if (FOLD && hi == (DH % 16) / 8) {
    Frag t;
    t.v = kf[DH / 16]; t.w[0] = 0x3F803F80u; t.w[1] = 0x00003F80u; kf[DH / 16] = t.v;
    t.v = vf[DH / 16]; t.w[0] = 0x3F803F80u; t.w[1] = 0x00003F80u; vf[DH / 16] = t.v;
  }

  f32x16 dvt[C::NDT], dkt[C::NDT];
#pragma unroll
  for (int dt = 0; dt < C::NDT; ++dt)
#pragma unroll
    for (int r = 0; r < 16; ++r) { dvt[dt][r] = 0.f; dkt[dt][r] = 0.f; }

  TileRegs<DH> qr, dor;
  float l_next = 0.f, d_next = 0.f;
  auto load_stats = [&](int q0) {
    if (threadIdx.x < 64) {
      const int qq = q0 + threadIdx.x;
      l_next = qq < p.T ? p.L[((long long)b * p.H + h) * p.T + qq] : INFINITY;
      d_next = qq < p.T ? p.Delta[((long long)b * p.H + h) * p.T + qq] : 0.f;
    }
  };
  constexpr bool PF = DH <= 80 && OCC < 3;  // register prefetch of the next tile (dh=160 would spill: load in place)
  const __amdgpu_buffer_rsrc_t rsQ = make_rsrc(Qb, (unsigned)(((long long)(p.T - 1) * p.ldq + DH) * 2));
  const __amdgpu_buffer_rsrc_t rsdO = make_rsrc(dOb, (unsigned)(((long long)(p.T - 1) * p.ldo + DH) * 2));
  qr.init(p.ldq); dor.init(p.ldo);
  if (PF) {
    qr.load(rsQ, t_begin);
    dor.load(rsdO, t_begin);
    load_stats(t_begin);
  }
  zero_pad_cols<DH>(Qs);
  zero_pad_cols<DH>(dOs);
  const float inv_scale2 = 1.f / p.scale2;
  for (int q0 = t_begin; q0 < t_end; q0 += 64) {
    if (!PF) {
      qr.load(rsQ, q0);
      dor.load(rsdO, q0);
      load_stats(q0);
    }
    __syncthreads();
    qr.store_rows(Qs);
    dor.store_rows(dOs);
    if (threadIdx.x < 64) {
      if (FOLD) {
        auto split3 = [](float x) {          // x ~ hi + mid + lo, each a bf16
          const bf16_t h0 = f2bf(x);
          const float r1 = x - bf2f(h0);
          const bf16_t h1 = f2bf(r1);
          const bf16_t h2 = f2bf(r1 - bf2f(h1));
          return make_uint4((uint32_t)h0 | ((uint32_t)h1 << 16), (uint32_t)h2, 0u, 0u);
        };
        // a query row beyond T has L = +inf: a large finite value keeps the pieces finite and still gives p = exp2(-huge) = 0
        *(uint4*)(Qs + img_off<C::LDE>(threadIdx.x, C::NCH)) = split3(fmaxf(-l_next * inv_scale2, -1e30f));
        *(uint4*)(dOs + img_off<C::LDE>(threadIdx.x, C::NCH)) = split3(-d_next);
      } else {
        Ls[threadIdx.x] = l_next; Dls[threadIdx.x] = d_next;
      }
    }
    __syncthreads();
    if (PF && q0 + 64 < t_end) {
      qr.load(rsQ, q0 + 64);
      dor.load(rsdO, q0 + 64);
      load_stats(q0 + 64);
    }
    const int nsub = (t_end - q0 > 32) ? 2 : 1;      // (chunks are multiples of 64 queries: only the last tile of T is ragged)
    // one 32-row sub-tile; at dh <= 48 the two sub-tiles are separate instantiations (fragment addresses become immediates,
    // no per-read address VALU), larger head dims keep the runtime loop (the unrolled form would spill)
    auto sub_tile = [&](auto sub_) __attribute__((always_inline)) {
      const int sub = sub_;
      f32x16 s, dp;
#pragma unroll
      for (int r = 0; r < 16; ++r) { s[r] = 0.f; dp[r] = 0.f; }
#pragma unroll
      for (int ks = 0; ks < C::NKS; ++ks) {
        s = probe_mfma(load_row_frag<DH>(Qs, fo, sub, ks), kf[ks], s);
        dp = probe_mfma(load_row_frag<DH>(dOs, fo, sub, ks), vf[ks], dp);
      }
      // No masks: a query row beyond T carries L = +inf (-> p = 0 exactly, dO row = 0 keeps dp finite), and a key lane
      // beyond S only pollutes its own accumulator column, which is never stored.
      float pr[16], ds[16];
      if (FOLD && ATTN_PROBE == 0) {
#pragma unroll
        for (int r = 0; r < 16; r += 2) {
          const f32x2 t = pk_mul<ATTN_PK_OLD>(f32x2{s[r], s[r + 1]}, f32x2{p.scale2, p.scale2});
          const f32x2 e2 = f32x2{fast_exp2(t.x), fast_exp2(t.y)};
          const f32x2 o2 = pk_mul<ATTN_PK_OLD>(e2, f32x2{dp[r], dp[r + 1]});
          pr[r] = e2.x; pr[r + 1] = e2.y; ds[r] = o2.x; ds[r + 1] = o2.y;
        }
      } else
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        const float4 l4 = *(const float4*)(Ls + sub * 32 + 8 * g + 4 * hi);     // acc rows 4g..4g+3 = 4 consecutive queries
        const float4 d4 = *(const float4*)(Dls + sub * 32 + 8 * g + 4 * hi);
        const float lq[4] = {l4.x, l4.y, l4.z, l4.w}, dq[4] = {d4.x, d4.y, d4.z, d4.w};
#pragma unroll
        for (int j = 0; j < 4; j += 2) {
          const int r = 4 * g + j;
          if (ATTN_PACKED && ATTN_PROBE == 0) {
            const f32x2 t = pk_fma<ATTN_PK_OLD>(f32x2{s[r], s[r + 1]}, f32x2{p.scale2, p.scale2}, f32x2{-lq[j], -lq[j + 1]});
            const f32x2 d = pk_fma<ATTN_PK_OLD>(f32x2{dq[j], dq[j + 1]}, f32x2{-1.f, -1.f}, f32x2{dp[r], dp[r + 1]});      // dp - dq, kept packed
            const f32x2 e2 = f32x2{fast_exp2(t.x), fast_exp2(t.y)};
            const f32x2 o2 = pk_mul<ATTN_PK_OLD>(e2, d);
            pr[r] = e2.x; pr[r + 1] = e2.y; ds[r] = o2.x; ds[r + 1] = o2.y;
          } else {
#pragma unroll
            for (int e = r; e < r + 2; ++e) {
              pr[e] = (ATTN_PROBE == 2 || ATTN_PROBE == 7) ? s[e] : ATTN_PROBE == 1 ? s[e] * p.scale2 - lq[e - 4 * g] : fast_exp2(s[e] * p.scale2 - lq[e - 4 * g]);
              ds[e] = (ATTN_PROBE == 2 || ATTN_PROBE == 7) ? dp[e] : pr[e] * (dp[e] - dq[e - 4 * g]);
            }
          }
        }
      }
      if (p.causal) {                      // (CLIP text encoder only; kept out of the unmasked loop above)
#pragma unroll
        for (int r = 0; r < 16; ++r)
          if (q0 + sub * 32 + acc_row(r, hi) < key) { pr[r] = 0.f; ds[r] = 0.f; }       // query before this lane's key
      }
      const bf16x8 pf0 = pack_acc(pr, 0), pf1 = pack_acc(pr, 1);
      const bf16x8 sf0 = pack_acc(ds, 0), sf1 = pack_acc(ds, 1);
#pragma unroll
      for (int dt = 0; dt < C::NDT; ++dt) {
        dvt[dt] = probe_mfma(load_T_frag<DH>(dOs, fo, dt, sub, 0), pf0, dvt[dt]);
        dvt[dt] = probe_mfma(load_T_frag<DH>(dOs, fo, dt, sub, 1), pf1, dvt[dt]);
        dkt[dt] = probe_mfma(load_T_frag<DH>(Qs, fo, dt, sub, 0), sf0, dkt[dt]);
        dkt[dt] = probe_mfma(load_T_frag<DH>(Qs, fo, dt, sub, 1), sf1, dkt[dt]);
      }
    };
    if constexpr (DH <= 48) {
      sub_tile(std::integral_constant<int, 0>{});
      if (nsub > 1) sub_tile(std::integral_constant<int, 1>{});
    } else {
#pragma unroll 1
      for (int sub = 0; sub < nsub; ++sub) sub_tile(sub);
    }
  }
  int krow = key;
  asm volatile("" : "+v"(krow));      // the output addresses are computed HERE (hoisted above the loop they cost the 168-register kernel a spill)
  if (p.tsplit > 1) {        // fp32 partials of this query chunk; the reduce kernel scales dK and rounds once
    float* base = p.part + ((((size_t)ts * gridDim.z + b) * p.H + h) * 2) * (size_t)p.S * DH;
    store_T_acc_f32<DH>(dvt, base, krow, p.S, hi);
    store_T_acc_f32<DH>(dkt, base + (size_t)p.S * DH, krow, p.S, hi);
    return;
  }
  store_T_acc<DH>(dvt, 1.f, p.dV + b * p.bv + h * DH, p.ldv, krow, p.S, hi);
  store_T_acc<DH>(dkt, p.scale, p.dK + b * p.bk + h * DH, p.ldk, krow, p.S, hi);
}

// ================================================================================================
// backward dK, dV for long query ranges with spare contraction slots (dh = 40), round 6: one wave = 64 keys, one wave per SIMD
// ================================================================================================
// The phased structure of attn_fwd64_kernel applied to the dK/dV kernel (same arithmetic as attn_bwd_dkv_kernel's FOLD path):
// a wave owns TWO 32-key blocks (K, V fragments in registers), so every Q / dO row fragment and every Q^T / dO^T transposed
// fragment read from LDS feeds two MFMAs, and a staged 64-query tile serves 256 keys.  A phase = the exp / dS VALU of one block
// (16 half-chunks of 3 instructions) with ONE of the other block's 14 MFMAs in front of each (its dV^T / dK^T updates of the
// previous sub-tile, then its S^T / dP^T of the next one).  14 MFMAs (448 clk) against ~270 clk of VALU: the phase is MFMA-bound.
// No branch inside the loop: padded query rows carry L = +inf in the folded column (p = 0), padded key lanes only pollute their own
// never-stored accumulator columns.  512 registers per lane (one workgroup per CU): 128 accumulators + 48 K/V fragments + 64
// score registers + the fragment sets of two sub-tiles.
__device__ __forceinline__ f32x2 pk_mul_s(f32x2 a, f32x2 sc) {
  f32x2 d;
  asm("v_pk_mul_f32 %0, %1, %2" : "=v"(d) : "v"(a), "s"(sc));
  return d;
}
// Score MFMAs of the one-wave-per-SIMD kernels: D in ARCH VGPRs (the softmax VALU reads it; left to the allocator under 400+
// live registers it lands in the accumulator file and comes back through 64 v_accvgpr_read per phase), B operand = a K / V / Q / dO
// fragment that lives in the ACCUMULATOR file for the whole kernel (MFMA reads A / B from either file).  Inline asm: the compiler
// pads no hazard here — every reader of D sits a whole phase (> 100 instructions) later, A comes from ds_read (counted by the
// compiler in front of the statement), B is written once in the prologue.
__device__ __forceinline__ void mfma_vab0(f32x16& d, bf16x8 a, bf16x8 b_acc) {
  asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, 0" : "=&v"(d) : "v"(a), "a"(b_acc));
}
__device__ __forceinline__ void mfma_vab(f32x16& d, bf16x8 a, bf16x8 b_acc) {
  asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, %0" : "+v"(d) : "v"(a), "a"(b_acc));
}
__device__ __forceinline__ uint4 split3_bf16(float x) {          // x ~ hi + mid + lo, each a bf16 (24 bits): {hi | mid << 16, lo, 0, 0}
  const bf16_t h0 = f2bf(x);
  const float r1 = x - bf2f(h0);
  const bf16_t h1 = f2bf(r1);
  const bf16_t h2 = f2bf(r1 - bf2f(h1));
  return make_uint4((uint32_t)h0 | ((uint32_t)h1 << 16), (uint32_t)h2, 0u, 0u);
}

// per-tile query statistics of the dK/dV kernel, one tile ahead in registers like the tile itself (row = thread & 63)
struct StatRegs {
  float lraw, draw;        // as loaded (row clamped to T - 1); the padded-row select happens at the point of USE, a tile later — applied
  int q;                   // at the load it would put the global-load latency of every tile in front of the next instruction
  __device__ __forceinline__ void load(const float* L, const float* Dl, int T) {      // q = this thread's query row of the tile being loaded
    const int qc = min(q, T - 1);
    lraw = L[qc]; draw = Dl[qc];
    q += 64;
  }
  __device__ __forceinline__ float l(int T) const { return q - 64 < T ? lraw : INFINITY; }
  __device__ __forceinline__ float d(int T) const { return q - 64 < T ? draw : 0.f; }
};

template <int DH, bool LOADS, int TOFF, int ROFF, bool BAR, int STAGE>
__device__ __forceinline__ void dkv64_phase(const float scale2, const float inv_scale2, const FragOff<DH>& fo, bf16_t* lds,
                                            const f32x16& sx, const f32x16& dpx, Frag (&pfx)[2], Frag (&sfx)[2],
                                            f32x16& sy, f32x16& dpy, const Frag (&pfy)[2], const Frag (&sfy)[2],
                                            f32x16 (&dvy)[Cfg<DH>::NDT], f32x16 (&dky)[Cfg<DH>::NDT],
                                            const bf16x8 (&kfy)[Cfg<DH>::NKS], const bf16x8 (&vfy)[Cfg<DH>::NKS],
                                            bf16x8 (&qT)[Cfg<DH>::NDT][2], bf16x8 (&doT)[Cfg<DH>::NDT][2],
                                            bf16x8 (&qrow)[Cfg<DH>::NKS], bf16x8 (&dorow)[Cfg<DH>::NKS],
                                            TileRegsV<DH>& qr, TileRegsV<DH>& dor, StatRegs& st, const __amdgpu_buffer_rsrc_t rsQ,
                                            const __amdgpu_buffer_rsrc_t rsdO, const float* Lb, const float* Db, const int T) {
  using C = Cfg<DH>;
  constexpr int NG = 4 * C::NDT, NM = NG + 2 * C::NKS, TILE = 64 * C::LDE;      // gradient MFMAs, then score MFMAs
  const f32x2 sc = {scale2, scale2};
  float e0 = 0.f, e1 = 0.f;
#pragma unroll
  for (int h = 0; h < 16; ++h) {
#pragma unroll
    for (int i = (h * NM + 15) / 16; i < ((h + 1) * NM + 15) / 16; ++i) {
      if (i < NG) {                                // dV^T / dK^T alternate, k-step outer: neighbours hit different accumulators
        const int w = i & 1, dt = (i >> 1) % C::NDT, k2 = (i >> 1) / C::NDT;
        if (w == 0) dvy[dt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(doT[dt][k2], pfy[k2].v, dvy[dt], 0, 0, 0);
        else dky[dt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(qT[dt][k2], sfy[k2].v, dky[dt], 0, 0, 0);
      } else {
        const int w = (i - NG) & 1, ks = (i - NG) >> 1;
        if (w == 0) { if (ks == 0) mfma_vab0(sy, qrow[ks], kfy[ks]); else mfma_vab(sy, qrow[ks], kfy[ks]); }
        else { if (ks == 0) mfma_vab0(dpy, dorow[ks], vfy[ks]); else mfma_vab(dpy, dorow[ks], vfy[ks]); }
      }
      A64_FENCE();
      if (LOADS && i == NG - 1 && ATTN64_PROBE != 4 && ATTN64_PROBE != 7) {                  // every transposed fragment has been consumed: fetch this sub-tile's
#pragma unroll
        for (int k2 = 0; k2 < 2; ++k2)
#pragma unroll
          for (int dt = 0; dt < C::NDT; ++dt) {
            const bf16_t* qb = lds + TOFF + k2 * 16 * C::LDE;
            qT[dt][k2] = tr_frag(qb + fo.tr_lo[dt], qb + fo.tr_hi[dt]);
            doT[dt][k2] = tr_frag(qb + TILE + fo.tr_lo[dt], qb + TILE + fo.tr_hi[dt]);
          }
        A64_FENCE();
      }
      if (LOADS && i == NM - 1 && BAR && ATTN64_PROBE != 6 && ATTN64_PROBE != 7) __syncthreads();
      if (LOADS && i == NM - 1 && (ATTN64_PROBE == 4 || ATTN64_PROBE == 7)) {
#pragma unroll
        for (int ks = 0; ks < C::NKS; ++ks) { asm volatile("" : "+v"(qrow[ks])); asm volatile("" : "+v"(dorow[ks])); }
      }
      if (LOADS && i == NM - 1 && ATTN64_PROBE != 4 && ATTN64_PROBE != 7) {                  // the next tile's images are complete; nobody reads the old tile any more
#pragma unroll
        for (int ks = 0; ks < C::NKS; ++ks) {
          qrow[ks] = *(const bf16x8*)(lds + ROFF + fo.row[ks]);
          dorow[ks] = *(const bf16x8*)(lds + ROFF + TILE + fo.row[ks]);
        }
        A64_FENCE();
      }
    }
    const int c = h >> 1;                          // score elements 2c, 2c + 1 of this lane's key
    if (ATTN64_PROBE == 2 || ATTN64_PROBE == 7) {
      if (h & 1) { pfx[c >> 2].w[c & 3] = __builtin_bit_cast(uint32_t, sx[2 * c]); sfx[c >> 2].w[c & 3] = __builtin_bit_cast(uint32_t, dpx[2 * c]); }
    } else if ((h & 1) == 0) {
      const f32x2 t = pk_mul_s(f32x2{sx[2 * c], sx[2 * c + 1]}, sc);
      e0 = fast_exp2(t.x); e1 = fast_exp2(t.y);
    } else {
      const f32x2 d2 = pk_mul<ATTN_PK_DMA>(f32x2{e0, e1}, f32x2{dpx[2 * c], dpx[2 * c + 1]});
      pfx[c >> 2].w[c & 3] = pack2bf(e0, e1);
      sfx[c >> 2].w[c & 3] = pack2bf(d2.x, d2.y);
    }
    A64_FENCE();
    if (STAGE >= 0 && ATTN64_PROBE != 5 && ATTN64_PROBE != 7) {
      if (h == 0) {
        qr.store_rows(lds + STAGE);
        *(uint4*)(lds + STAGE + img_off<C::LDE>(threadIdx.x & 63, C::NCH)) = split3_bf16(fmaxf(-st.l(T) * inv_scale2, -1e30f));
        A64_FENCE();
      }
      if (h == 2) {
        dor.store_rows(lds + STAGE + TILE);
        *(uint4*)(lds + STAGE + TILE + img_off<C::LDE>(threadIdx.x & 63, C::NCH)) = split3_bf16(-st.d(T));
        A64_FENCE();
      }
      if (h == 4) { qr.load(rsQ); A64_FENCE(); }
      if (h == 6) { dor.load(rsdO); st.load(Lb, Db, T); A64_FENCE(); }
    }
  }
}

template <int DH>
__global__ __launch_bounds__(256, 1) void attn_bwd_dkv64_kernel(AttnArgs p) {
  using C = Cfg<DH>;
  static_assert(DH % 16 == 8 && C::DK - DH >= 3, "needs three spare contraction slots (folded L / Delta)");
  constexpr int TILE = 64 * C::LDE;
  __shared__ __attribute__((aligned(16))) bf16_t lds[4 * TILE];       // [Q image 0][dO image 0][Q image 1][dO image 1]
  const FragOff<DH> fo;
  const Blk blk = xcd_block(p.xcd_raster);
  const int b = blk.b, h = blk.h;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int li = lane & 31, hi = lane >> 5;
  const int key0 = blk.x * 256 + wave * 64 + li;                       // block 0: key0, block 1: key0 + 32
  const bf16_t* Qb = p.Q + b * p.bq + h * DH;
  const bf16_t* dOb = p.dO + b * p.bo + h * DH;
  const float* Lb = p.L + ((long long)b * p.H + h) * p.T;
  const float* Db = p.Delta + ((long long)b * p.H + h) * p.T;

  bf16x8 kf[2][C::NKS], vf[2][C::NKS];
#pragma unroll
  for (int x = 0; x < 2; ++x) {
    load_row_frags<DH>(p.K + b * p.bk + h * DH, p.ldk, key0 + 32 * x, p.S, hi, kf[x]);
    load_row_frags<DH>(p.V + b * p.bv + h * DH, p.ldv, key0 + 32 * x, p.S, hi, vf[x]);
    if (hi == 1) {                                 // ones against the three folded bf16 pieces of -L / scale2 (Q image) and -Delta (dO image)
      Frag t;
      t.q = make_uint4(0x3F803F80u, 0x00003F80u, 0u, 0u);
      kf[x][DH / 16] = t.v; vf[x][DH / 16] = t.v;
    }
  }
  const float inv_scale2 = 1.f / p.scale2;
  TileRegsV<DH> qr, dor;
  StatRegs st;
  st.q = threadIdx.x & 63;
  const __amdgpu_buffer_rsrc_t rsQ = make_rsrc(Qb, (unsigned)(((long long)(p.T - 1) * p.ldq + DH) * 2));
  const __amdgpu_buffer_rsrc_t rsdO = make_rsrc(dOb, (unsigned)(((long long)(p.T - 1) * p.ldo + DH) * 2));
  qr.init(p.ldq); dor.init(p.ldo);
#pragma unroll
  for (int tb = 0; tb < 2; ++tb) {                 // tiles 0 and 1 -> the two buffers
    qr.load(rsQ); dor.load(rsdO); st.load(Lb, Db, p.T);
    qr.store_rows(lds + tb * 2 * TILE); dor.store_rows(lds + tb * 2 * TILE + TILE);
    *(uint4*)(lds + tb * 2 * TILE + img_off<C::LDE>(threadIdx.x & 63, C::NCH)) = split3_bf16(fmaxf(-st.l(p.T) * inv_scale2, -1e30f));
    *(uint4*)(lds + tb * 2 * TILE + TILE + img_off<C::LDE>(threadIdx.x & 63, C::NCH)) = split3_bf16(-st.d(p.T));
  }
  qr.load(rsQ); dor.load(rsdO); st.load(Lb, Db, p.T);                   // tile 2 waits in registers for the first odd X phase

  f32x16 dv0[C::NDT], dk0[C::NDT], dv1[C::NDT], dk1[C::NDT], s0, dp0, s1, dp1;
  Frag pf0[2], sf0[2], pf1[2], sf1[2];
  bf16x8 qT[C::NDT][2], doT[C::NDT][2], qrow[C::NKS], dorow[C::NKS];
#pragma unroll
  for (int dt = 0; dt < C::NDT; ++dt)
#pragma unroll
    for (int r = 0; r < 16; ++r) { dv0[dt][r] = 0.f; dk0[dt][r] = 0.f; dv1[dt][r] = 0.f; dk1[dt][r] = 0.f; }
  {
    Frag z;
    z.q = make_uint4(0u, 0u, 0u, 0u);
    pf0[0] = z; pf0[1] = z; sf0[0] = z; sf0[1] = z; pf1[0] = z; pf1[1] = z; sf1[0] = z; sf1[1] = z;
#pragma unroll
    for (int dt = 0; dt < C::NDT; ++dt) { qT[dt][0] = z.v; qT[dt][1] = z.v; doT[dt][0] = z.v; doT[dt][1] = z.v; }
  }
  __syncthreads();
#pragma unroll
  for (int ks = 0; ks < C::NKS; ++ks) {
    qrow[ks] = *(const bf16x8*)(lds + fo.row[ks]);
    dorow[ks] = *(const bf16x8*)(lds + TILE + fo.row[ks]);
  }
#pragma unroll
  for (int r = 0; r < 16; ++r) { s0[r] = 0.f; dp0[r] = 0.f; }
#pragma unroll
  for (int ks = 0; ks < C::NKS; ++ks) {
    mfma_vab(s0, qrow[ks], kf[0][ks]);
    mfma_vab(dp0, dorow[ks], vf[0][ks]);
  }
  asm volatile("s_nop 7\n\ts_nop 7" ::: "memory");          // (these D registers are read a few instructions into the first phase)

  const int ntiles = (p.T + 63) >> 6;
  auto tile = [&](auto tb_) __attribute__((always_inline)) {
    constexpr int TB = decltype(tb_)::value;
    constexpr int QI = TB * 2 * TILE, QN = (1 - TB) * 2 * TILE;
#define DKV64_ARGS_Y s0, dp0, pf0, sf0, s1, dp1, pf1, sf1, dv1, dk1, kf[1], vf[1]
#define DKV64_ARGS_X s1, dp1, pf1, sf1, s0, dp0, pf0, sf0, dv0, dk0, kf[0], vf[0]
#define DKV64_TAIL qT, doT, qrow, dorow, qr, dor, st, rsQ, rsdO, Lb, Db, p.T
    dkv64_phase<DH, true, QI, QI + 32 * C::LDE, false, -1>(p.scale2, inv_scale2, fo, lds, DKV64_ARGS_Y, DKV64_TAIL);
    dkv64_phase<DH, false, 0, 0, false, -1>(p.scale2, inv_scale2, fo, lds, DKV64_ARGS_X, DKV64_TAIL);
    dkv64_phase<DH, true, QI + 32 * C::LDE, QN, true, -1>(p.scale2, inv_scale2, fo, lds, DKV64_ARGS_Y, DKV64_TAIL);
    dkv64_phase<DH, false, 0, 0, false, QI>(p.scale2, inv_scale2, fo, lds, DKV64_ARGS_X, DKV64_TAIL);
#undef DKV64_ARGS_Y
#undef DKV64_ARGS_X
#undef DKV64_TAIL
  };
  for (int t = 0; t < ntiles; t += 2) {
    tile(std::integral_constant<int, 0>{});
    if (t + 1 < ntiles) tile(std::integral_constant<int, 1>{});
  }
#pragma unroll
  for (int k2 = 0; k2 < 2; ++k2)                   // the last sub-tile's gradient updates of block 1
#pragma unroll
    for (int dt = 0; dt < C::NDT; ++dt) {
      dv1[dt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(doT[dt][k2], pf1[k2].v, dv1[dt], 0, 0, 0);
      dk1[dt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(qT[dt][k2], sf1[k2].v, dk1[dt], 0, 0, 0);
    }
  int krow = key0;
  asm volatile("" : "+v"(krow));
  bf16_t* dVb = p.dV + b * p.bv + h * DH;
  bf16_t* dKb = p.dK + b * p.bk + h * DH;
  store_T_acc<DH>(dv0, 1.f, dVb, p.ldv, krow, p.S, hi);
  store_T_acc<DH>(dk0, p.scale, dKb, p.ldk, krow, p.S, hi);
  store_T_acc<DH>(dv1, 1.f, dVb, p.ldv, krow + 32, p.S, hi);
  store_T_acc<DH>(dk1, p.scale, dKb, p.ldk, krow + 32, p.S, hi);
}

// ================================================================================================
// backward dK, dV with LDS-DMA tile staging (round 6, dh = 40, un-split query range): attn_bwd_dkv_kernel's arithmetic and occupancy
// (32 keys per wave, three workgroups per CU, FOLD), but the Q / dO tiles arrive by buffer_load ... lds into a ring of three
// compact (128-byte-row) tile buffers, two tiles ahead of their use: no staging registers, no ds_write of the tiles, no global-load
// latency in front of the stores, ONE barrier per 64-query tile instead of two.
//   iteration t:  vmcnt(6) [tile t's pieces landed; tile t + 1's and its statistics stay in flight] -> pad columns of tile t
//                 (folded -L / scale2, -Delta: from registers loaded two iterations ago) -> barrier -> statistics + DMA of tile t + 2
//                 into the buffer of tile t - 1 -> the two 32-query sub-tiles of tile t.
// Every wave issues the same VMEM sequence (2 statistics loads, 4 DMA pieces per tile), so the counted waits hold on every wave.
// ================================================================================================
#ifndef ATTN_DKV_DMA
#define ATTN_DKV_DMA 1
#endif

template <int DH>
__global__ __launch_bounds__(256, 3) void attn_bwd_dkv_dma_kernel(AttnArgs p) {
  using C = Cfg<DH>;
  static_assert(DH % 16 == 8 && C::DK - DH >= 3 && C::DV <= 64, "folded statistics need three spare contraction slots; compact images 64 columns");
  constexpr int IMG = 64 * 64, STAT = 512, TILE = 2 * IMG + STAT, NB = 3;           // slot = [Q image 8 KB][dO image 8 KB][{L, Delta} x 64 + slack: 1 KB]
  __shared__ __attribute__((aligned(1024))) bf16_t lds[NB * TILE];                  // 51 KB: three workgroups per CU
  const FragOff64<DH> fo;
  const Blk blk = xcd_block(p.xcd_raster);
  const int b = blk.b, h = blk.h;
  const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int li = lane & 31, hi = lane >> 5;
  const int key = blk.x * 128 + wave * 32 + li;
  const bf16_t* Qb = p.Q + b * p.bq + h * DH;
  const bf16_t* dOb = p.dO + b * p.bo + h * DH;
  const float* LDb = p.LD + 2 * ((long long)b * p.H + h) * p.T;

  bf16x8 kf[C::NKS], vf[C::NKS];
  load_row_frags<DH>(p.K + b * p.bk + h * DH, p.ldk, key, p.S, hi, kf);
  load_row_frags<DH>(p.V + b * p.bv + h * DH, p.ldv, key, p.S, hi, vf);
  if (hi == 1) {                                   // ones against the three folded bf16 pieces (see attn_bwd_dkv_kernel)
    Frag t;
    t.q = make_uint4(0x3F803F80u, 0x00003F80u, 0u, 0u);
    kf[DH / 16] = t.v; vf[DH / 16] = t.v;
  }
  f32x16 dvt[C::NDT], dkt[C::NDT];
#pragma unroll
  for (int dt = 0; dt < C::NDT; ++dt)
#pragma unroll
    for (int r = 0; r < 16; ++r) { dvt[dt][r] = 0.f; dkt[dt][r] = 0.f; }

  TileDma<DH> qd, dod;
  const DmaRsrc rsQ = make_dma_rsrc(Qb, (unsigned)(((long long)(p.T - 1) * p.ldq + DH) * 2));
  const DmaRsrc rsdO = make_dma_rsrc(dOb, (unsigned)(((long long)(p.T - 1) * p.ldo + DH) * 2));
  const DmaRsrc rsLD = make_dma_rsrc(LDb, (unsigned)(p.T * 8));
  qd.init(p.ldq); dod.init(p.ldo);
  // Every wave also fetches the {L, Delta} pairs of ITS 16 query rows of the tile (rows [16 wave, +16): the rows its own tile pieces
  // cover) as one 4-bytes-per-lane piece — 32 floats = lanes 0..31, the other lanes out of range — into its own 256 bytes of the
  // slot's statistics area, and turns them into the folded pad columns of exactly those rows behind its own vmcnt wait (a pad written
  // into a row whose piece another wave still has in flight would be zero-filled again when that piece lands).  NOTHING in the loop is
  // a compiler-counted VMEM load (see dma16), so the counted waits are exact: 5 pieces per tile and wave.
  unsigned svo = lane < 32 ? (unsigned)((32 * wave + lane) * 4) : 0x80000000u;
  const unsigned sstep = lane < 32 ? 512u : 0u;
  auto issue_tile = [&](bf16_t* slot) {
    qd.issue(rsQ, slot); dod.issue(rsdO, slot + IMG);
    dma4(rsLD, svo, __builtin_amdgcn_readfirstlane(lds_addr(slot + 2 * IMG) + wave * 256));
    svo += sstep;
  };
  issue_tile(lds);                                 // tile 0
  issue_tile(lds + TILE);                          // tile 1
  const float inv_scale2 = 1.f / p.scale2;
  const int ntiles = (p.T + 63) >> 6;

  auto tile = [&](auto tb_, int t) __attribute__((always_inline)) {
    constexpr int TB = decltype(tb_)::value;
    constexpr int QI = TB * TILE, DI = QI + IMG, SI = QI + 2 * IMG, NXT = ((TB + 2) % NB) * TILE;      // tile t + 2 goes where tile t - 1 was
    attn_wait_vmcnt<5>();                          // tile t's pieces (and this wave's statistics) have landed; tile t + 1's stay in flight
    if (lane < 32) {                               // lanes 0..15: Q image pad of row 16 wave + lane; 16..31: dO image pad of the same rows
      const int r = 16 * wave + (lane & 15);
      const float2 ld = *(const float2*)((const float*)(lds + SI) + 64 * wave + 2 * (lane & 15));
      const int q = t * 64 + r;
      const float x = lane < 16 ? fmaxf(-(q < p.T ? ld.x : INFINITY) * inv_scale2, -1e30f) : (q < p.T ? -ld.y : 0.f);
      *(uint4*)(lds + (lane < 16 ? QI : DI) + img64_off(r, C::NCH)) = split3_bf16(x);
    }
    __syncthreads();
    issue_tile(lds + NXT);
    auto sub_tile = [&](auto sub_) __attribute__((always_inline)) {
      constexpr int SO = decltype(sub_)::value * 32 * 64;
      f32x16 s, dp;
#pragma unroll
      for (int r = 0; r < 16; ++r) { s[r] = 0.f; dp[r] = 0.f; }
#pragma unroll
      for (int ks = 0; ks < C::NKS; ++ks) {
        s = __builtin_amdgcn_mfma_f32_32x32x16_bf16(*(const bf16x8*)(lds + QI + SO + fo.row[ks]), kf[ks], s, 0, 0, 0);
        dp = __builtin_amdgcn_mfma_f32_32x32x16_bf16(*(const bf16x8*)(lds + DI + SO + fo.row[ks]), vf[ks], dp, 0, 0, 0);
      }
      float pr[16], ds[16];
#pragma unroll
      for (int r = 0; r < 16; r += 2) {
        const f32x2 t2 = pk_mul<ATTN_PK_DMA>(f32x2{s[r], s[r + 1]}, f32x2{p.scale2, p.scale2});
        const f32x2 e2 = f32x2{fast_exp2(t2.x), fast_exp2(t2.y)};
        const f32x2 o2 = pk_mul<ATTN_PK_DMA>(e2, f32x2{dp[r], dp[r + 1]});
        pr[r] = e2.x; pr[r + 1] = e2.y; ds[r] = o2.x; ds[r + 1] = o2.y;
      }
      const bf16x8 pf0 = pack_acc(pr, 0), pf1 = pack_acc(pr, 1);
      const bf16x8 sf0 = pack_acc(ds, 0), sf1 = pack_acc(ds, 1);
#pragma unroll
      for (int dt = 0; dt < C::NDT; ++dt) {
        const bf16_t* q0 = lds + QI + SO, *q1 = q0 + 16 * 64, *o0 = lds + DI + SO, *o1 = o0 + 16 * 64;
        dvt[dt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(tr_frag(o0 + fo.tr_lo[dt], o0 + fo.tr_hi[dt]), pf0, dvt[dt], 0, 0, 0);
        dvt[dt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(tr_frag(o1 + fo.tr_lo[dt], o1 + fo.tr_hi[dt]), pf1, dvt[dt], 0, 0, 0);
        dkt[dt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(tr_frag(q0 + fo.tr_lo[dt], q0 + fo.tr_hi[dt]), sf0, dkt[dt], 0, 0, 0);
        dkt[dt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(tr_frag(q1 + fo.tr_lo[dt], q1 + fo.tr_hi[dt]), sf1, dkt[dt], 0, 0, 0);
      }
    };
    sub_tile(std::integral_constant<int, 0>{});
    if (p.T - t * 64 > 32) sub_tile(std::integral_constant<int, 1>{});
  };
  for (int t = 0; t < ntiles; t += 3) {
    tile(std::integral_constant<int, 0>{}, t);
    if (t + 1 < ntiles) tile(std::integral_constant<int, 1>{}, t + 1);
    if (t + 2 < ntiles) tile(std::integral_constant<int, 2>{}, t + 2);
  }
  attn_wait_vmcnt<0>();                           // (DMA pieces of tiles beyond the end must not outlive the workgroup's LDS)
  int krow = key;
  asm volatile("" : "+v"(krow));
  store_T_acc<DH>(dvt, 1.f, p.dV + b * p.bv + h * DH, p.ldv, krow, p.S, hi);
  store_T_acc<DH>(dkt, p.scale, p.dK + b * p.bk + h * DH, p.ldk, krow, p.S, hi);
}

// dV / dK = sum over the query chunks' partials (fixed order: deterministic), dK scaled, one bf16 rounding
template <int DH>
__global__ __launch_bounds__(256) void attn_dkv_reduce_kernel(AttnArgs p, int Bn) {
  constexpr int D4 = DH / 4;
  const long long per = (long long)p.S * D4, total = (long long)Bn * p.H * 2 * per;
  const size_t slab = (size_t)Bn * p.H * 2 * p.S * DH;
  for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long long)gridDim.x * 256) {
    const long long bh2 = i / per;
    const int rem = (int)(i - bh2 * per), key = rem / D4, d = (rem - key * D4) * 4;
    const int which = (int)(bh2 & 1), h = (int)((bh2 >> 1) % p.H), b = (int)((bh2 >> 1) / p.H);
    const float* src = p.part + (size_t)bh2 * p.S * DH + (size_t)key * DH + d;
    float4 a = *(const float4*)src;
    for (int t = 1; t < p.tsplit; ++t) {
      const float4 v = *(const float4*)(src + (size_t)t * slab);
      a.x += v.x; a.y += v.y; a.z += v.z; a.w += v.w;
    }
    const float mul = which ? p.scale : 1.f;
    const float f[4] = {a.x * mul, a.y * mul, a.z * mul, a.w * mul};
    bf16_t* dst = which ? p.dK + b * p.bk + (long long)key * p.ldk + h * DH + d : p.dV + b * p.bv + (long long)key * p.ldv + h * DH + d;
    *(uint2*)dst = pack4(f);
  }
}

inline int xcd_raster_on() {                      // (the hardware's round-robin deal measured equal in time at 3.1-4.6 x the HBM traffic: profiles/r03_attention_xcd_raster_checks.txt)
  static const int on = 1;
  return on;
}

// Query chunks of the dK/dV kernel.  It parallelises over (key block, head, batch) and walks the queries serially: with S = 77
// (cross-attention: ONE key block) that is B * H = 128 workgroups — half the chip — each making 64 trips of 64 queries at the
// 64 x 64 level, 120 us of pure loop latency per launch (0.095 of the HBM roof, profiles/r03_roofline_per_shape.csv).  Cutting T
// so that ~1024 workgroups exist turns it into ~8 trips each; the partials are tiny (S x DH per head).  Long key ranges
// (self-attention) already fill the chip: never split.
inline void dkv_tsplit(int Bn, int H, int T, int S, int* tsplit, int* tchunk) {
  const bool off = false;
  const long long wgs = (long long)cdiv(S, 128) * H * Bn;
  int n = 1;
  if (!off && wgs < 512 && T >= 512) {
    n = (int)((1024 + wgs - 1) / wgs);
    if (n > T / 256) n = T / 256;
    if (n < 1) n = 1;
  }
  const int chunk = cdiv(cdiv(T, n), 64) * 64;
  *tchunk = chunk;
  *tsplit = cdiv(T, chunk);
}

template <int DH>
int launch_fwd(const AttnArgs& p, int Bn, hipStream_t st) {
  // algorithmic bytes: Q, K, V read once, O written once (bf16) + the fp32 log-sum-exp
  const bool fwd64 = DH == 40 && ATTN_FWD64 && !p.causal && p.S >= ATTN_FWD64_MIN_S;
  E4T_LOG_LAUNCH("%s<%d>|B%d H%d T%d S%d causal%d|%.0f|%.0f", fwd64 ? "attn_fwd64_kernel" : "attn_fwd_kernel", DH, Bn, p.H, p.T, p.S, p.causal,
                 2.0 * Bn * p.H * DH * (2.0 * p.T + 2.0 * p.S) + 4.0 * Bn * p.H * p.T, 4.0 * Bn * p.H * (double)p.T * p.S * DH);
  if constexpr (DH == 40) {
    if (fwd64) {
      hipLaunchKernelGGL((attn_fwd64_kernel<DH, ATTN_FWD64_ROWS>), dim3(cdiv(p.T, 256), p.H, Bn), dim3(256), 0, st, p);
      E4T_CHECK_LAUNCH("attn_fwd64_kernel");
      return 0;
    }
  }
  const int probe_lds = 0;
  hipLaunchKernelGGL((attn_fwd_kernel<DH>), dim3(cdiv(p.T, 128), p.H, Bn), dim3(256), probe_lds, st, p);
  E4T_CHECK_LAUNCH("attn_fwd_kernel");
  return 0;
}
template <int DH>
int launch_bwd(AttnArgs p, int Bn, size_t ws_floats, hipStream_t st) {
  const long long total = (long long)Bn * p.H * p.T;
  dkv_tsplit(Bn, p.H, p.T, p.S, &p.tsplit, &p.tchunk);
  if (p.tsplit > 1) {
    const size_t need = (size_t)total + (size_t)p.tsplit * Bn * p.H * 2 * p.S * DH;
    if (ws_floats >= need) p.part = p.Delta + (((size_t)total + 3) & ~(size_t)3);     // 16-byte aligned behind Delta
    if (ws_floats < need + 3) { p.tsplit = 1; p.part = nullptr; }                     // caller sized the workspace for Delta only
  } else if (ws_floats >= 3 * (size_t)total + 4) {
    p.LD = p.Delta + (((size_t)total + 3) & ~(size_t)3);                              // {L, Delta} pairs for the DMA-staged dK/dV kernel
  }
  // measured (tools/ab_dkv.py, dh 40, B16 H8 T4096): S = 4096 1.790 vs 1.835 ms per backward with 3 workgroups per CU, S = 77
  // 0.194 vs 0.167 ms (one workgroup per (batch, head): nothing to cover the un-prefetched tile loads) -> long key ranges only
  // dh 64 (SD-2.x): three workgroups per CU cost the kernel a 16-byte spill and buy nothing (C5 B = 4: 85.3 vs 85.8 ms per step, B = 1 equal;
  // profiles/r04_ab/r04g_c5_occ*): it stays at two
  const int dkv_occ = DH > 64 ? 1 : ((p.S >= 2048 && DH < 64) ? 3 : DKV_WAVES);
  const bool dkv64 = DH == 40 && ATTN_BWD64 && !p.causal && p.tsplit == 1 && p.T >= ATTN_FWD64_MIN_S && p.S >= 256;
  const bool dq_dma = DH <= 64 && ATTN_DQ_DMA && !p.causal && p.S >= 192;      // (dh 32 / 40 / 64: rows of at most 128 bytes)
  const bool dkv_dma = DH == 40 && ATTN_DKV_DMA && !dkv64 && !p.causal && p.tsplit == 1 && p.T >= 192 && p.LD != nullptr && (long long)p.T * 8 < 0x7fffffffLL;
  if (e4t_launch_log_enabled()) {
    const double el = (double)Bn * p.H * DH;      // elements per token row over all heads
    E4T_LOG_LAUNCH("%s<%d>|B%d H%d T%d S%d causal%d|%.0f|%.0f", dq_dma ? "attn_bwd_dq_dma_kernel" : "attn_bwd_dq_kernel", DH, Bn, p.H, p.T, p.S, p.causal,
                   2.0 * el * (4.0 * p.T + 2.0 * p.S) + 8.0 * Bn * p.H * p.T, 6.0 * Bn * p.H * (double)p.T * p.S * DH);
    if (dkv_dma)
      E4T_LOG_LAUNCH("attn_bwd_dkv_dma_kernel<%d>|B%d H%d T%d S%d causal%d|%.0f|%.0f", DH, Bn, p.H, p.T, p.S, p.causal,
                     2.0 * el * (2.0 * p.T + 4.0 * p.S) + 8.0 * Bn * p.H * p.T, 8.0 * Bn * p.H * (double)p.T * p.S * DH);
    else if (dkv64)
      E4T_LOG_LAUNCH("attn_bwd_dkv64_kernel<%d>|B%d H%d T%d S%d causal%d|%.0f|%.0f", DH, Bn, p.H, p.T, p.S, p.causal,
                     2.0 * el * (2.0 * p.T + 4.0 * p.S) + 8.0 * Bn * p.H * p.T, 8.0 * Bn * p.H * (double)p.T * p.S * DH);
    else
      E4T_LOG_LAUNCH("attn_bwd_dkv_kernel<%d, %d>|B%d H%d T%d S%d causal%d|%.0f|%.0f", DH, dkv_occ, Bn, p.H, p.T, p.S, p.causal,
                     2.0 * el * (2.0 * p.T + 4.0 * p.S) + 8.0 * Bn * p.H * p.T, 8.0 * Bn * p.H * (double)p.T * p.S * DH);
  }
  // dQ first: its prologue also produces Delta (row_delta), which the dK/dV kernel behind it reads
  bool dq_done = false;
  if constexpr (DH <= 64) {
    if (dq_dma) {
      hipLaunchKernelGGL((attn_bwd_dq_dma_kernel<DH>), dim3(cdiv(p.T, 128), p.H, Bn), dim3(256), 0, st, p);
      E4T_CHECK_LAUNCH("attn_bwd_dq_dma_kernel");
      dq_done = true;
    }
  }
  if (!dq_done) {
    hipLaunchKernelGGL((attn_bwd_dq_kernel<DH>), dim3(cdiv(p.T, 128), p.H, Bn), dim3(256), 0, st, p);
    E4T_CHECK_LAUNCH("attn_bwd_dq_kernel");
  }
  const dim3 gdkv(cdiv(p.S, 128) * p.tsplit, p.H, Bn);
  if constexpr (DH == 40) {
    if (dkv_dma) {
      hipLaunchKernelGGL((attn_bwd_dkv_dma_kernel<DH>), gdkv, dim3(256), 0, st, p);
      E4T_CHECK_LAUNCH("attn_bwd_dkv_dma_kernel");
      return 0;
    }
    if (dkv64) {
      hipLaunchKernelGGL((attn_bwd_dkv64_kernel<DH>), dim3(cdiv(p.S, 256), p.H, Bn), dim3(256), 0, st, p);
      E4T_CHECK_LAUNCH("attn_bwd_dkv64_kernel");
      return 0;
    }
  }
  if constexpr (DH <= 64) {
    if (dkv_occ == 3) hipLaunchKernelGGL((attn_bwd_dkv_kernel<DH, 3>), gdkv, dim3(256), 0, st, p);
    else hipLaunchKernelGGL((attn_bwd_dkv_kernel<DH, DKV_WAVES>), gdkv, dim3(256), 0, st, p);
  } else {
    hipLaunchKernelGGL((attn_bwd_dkv_kernel<DH, 1>), gdkv, dim3(256), 0, st, p);
  }
  E4T_CHECK_LAUNCH("attn_bwd_dkv_kernel");
  if (p.tsplit > 1) {
    const long long items = (long long)Bn * p.H * 2 * p.S * (DH / 4);
    int rb = (int)((items + 255) / 256);
    if (rb > 2048) rb = 2048;
    hipLaunchKernelGGL((attn_dkv_reduce_kernel<DH>), dim3(rb), dim3(256), 0, st, p, Bn);
    E4T_CHECK_LAUNCH("attn_dkv_reduce_kernel");
  }
  return 0;
}

int check_common(int Bn, int H, int T, int S, int DH, int ldq, int ldk, int ldv, int ldo) {
  E4T_REQUIRE(Bn > 0 && H > 0 && T > 0 && S > 0, "attention: bad shape B=%d H=%d T=%d S=%d", Bn, H, T, S);
  E4T_REQUIRE(DH == 40 || DH == 64 || DH == 80 || DH == 160 || DH == 32, "attention: head dim %d not built (32/40/64/80/160)", DH);
  E4T_REQUIRE(ldq % 8 == 0 && ldk % 8 == 0 && ldv % 8 == 0 && ldo % 8 == 0, "attention: row strides must be multiples of 8");
  return 0;
}

}  // namespace

extern "C" int e4t_attention_fwd(const void* Q, const void* K, const void* V, void* O, float* lse, int Bn, int H, int T, int S,
                                 int DH, int ldq, int ldk, int ldv, int ldo, long long bq, long long bk, long long bv,
                                 long long bo, float scale, int causal, e4t_stream stream) {
  if (int e = check_common(Bn, H, T, S, DH, ldq, ldk, ldv, ldo)) return e;
  E4T_REQUIRE(Q && K && V && O, "attention_fwd: null operand");
  AttnArgs p;
  memset(&p, 0, sizeof(p));
  p.Q = (const bf16_t*)Q; p.K = (const bf16_t*)K; p.V = (const bf16_t*)V; p.Out = (bf16_t*)O; p.L = lse;
  p.T = T; p.S = S; p.H = H; p.ldq = ldq; p.ldk = ldk; p.ldv = ldv; p.ldo = ldo; p.bq = bq; p.bk = bk; p.bv = bv; p.bo = bo;
  p.scale = scale; p.scale2 = scale * 1.4426950408889634f; p.causal = causal; p.xcd_raster = xcd_raster_on();
  hipStream_t st = (hipStream_t)stream;
  switch (DH) {
    case 32: return launch_fwd<32>(p, Bn, st);
    case 40: return launch_fwd<40>(p, Bn, st);
    case 64: return launch_fwd<64>(p, Bn, st);
    case 80: return launch_fwd<80>(p, Bn, st);
    default: return launch_fwd<160>(p, Bn, st);
  }
}

extern "C" size_t e4t_attention_bwd_workspace_floats(int Bn, int H, int T, int S, int DH) {
  if (Bn <= 0 || H <= 0 || T <= 0 || S <= 0 || DH <= 0) return 0;
  int tsplit, tchunk;
  dkv_tsplit(Bn, H, T, S, &tsplit, &tchunk);
  const size_t delta = (size_t)Bn * H * T;
  return tsplit > 1 ? delta + 4 + (size_t)tsplit * Bn * H * 2 * S * DH : 3 * delta + 4;      // un-split: Delta + the {L, Delta} pairs
}

extern "C" int e4t_attention_bwd(const void* Q, const void* K, const void* V, const void* O, const void* dO, const float* lse,
                                 float* delta_ws, void* dQ, void* dK, void* dV, int Bn, int H, int T, int S, int DH, int ldq,
                                 int ldk, int ldv, int ldo, long long bq, long long bk, long long bv, long long bo, float scale,
                                 int causal, e4t_stream stream) {
  return e4t_attention_bwd_ws(Q, K, V, O, dO, lse, delta_ws, (size_t)(Bn > 0 && H > 0 && T > 0 ? (size_t)Bn * H * T : 0), dQ, dK, dV, Bn, H, T, S, DH,
                              ldq, ldk, ldv, ldo, bq, bk, bv, bo, scale, causal, stream);
}

extern "C" int e4t_attention_bwd_ws(const void* Q, const void* K, const void* V, const void* O, const void* dO, const float* lse,
                                    float* delta_ws, size_t ws_floats, void* dQ, void* dK, void* dV, int Bn, int H, int T, int S, int DH,
                                    int ldq, int ldk, int ldv, int ldo, long long bq, long long bk, long long bv, long long bo, float scale,
                                    int causal, e4t_stream stream) {
  if (int e = check_common(Bn, H, T, S, DH, ldq, ldk, ldv, ldo)) return e;
  E4T_REQUIRE(Q && K && V && O && dO && lse && delta_ws && dQ && dK && dV, "attention_bwd: null operand");
  E4T_REQUIRE(ws_floats >= (size_t)Bn * H * T, "attention_bwd: workspace of %zu floats is smaller than B*H*T", ws_floats);
  E4T_REQUIRE(((uintptr_t)delta_ws & 15) == 0, "attention_bwd: workspace must be 16-byte aligned");
  AttnArgs p;
  memset(&p, 0, sizeof(p));
  p.Q = (const bf16_t*)Q; p.K = (const bf16_t*)K; p.V = (const bf16_t*)V; p.O = (const bf16_t*)O; p.dO = (const bf16_t*)dO;
  p.L = (float*)lse; p.Delta = delta_ws; p.dQ = (bf16_t*)dQ; p.dK = (bf16_t*)dK; p.dV = (bf16_t*)dV;
  p.T = T; p.S = S; p.H = H; p.ldq = ldq; p.ldk = ldk; p.ldv = ldv; p.ldo = ldo; p.bq = bq; p.bk = bk; p.bv = bv; p.bo = bo;
  p.scale = scale; p.scale2 = scale * 1.4426950408889634f; p.causal = causal; p.xcd_raster = xcd_raster_on();
  hipStream_t st = (hipStream_t)stream;
  switch (DH) {
    case 32: return launch_bwd<32>(p, Bn, ws_floats, st);
    case 40: return launch_bwd<40>(p, Bn, ws_floats, st);
    case 64: return launch_bwd<64>(p, Bn, ws_floats, st);
    case 80: return launch_bwd<80>(p, Bn, ws_floats, st);
    default: return launch_bwd<160>(p, Bn, ws_floats, st);
  }
}
