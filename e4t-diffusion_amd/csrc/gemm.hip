// MFMA GEMM / implicit-GEMM 3x3 convolution for gfx950 (CDNA4).
//
//   C[M,N] = epilogue( alpha * A[M,K] . B[N,K]^T )          ("NT": both operands K-contiguous)
//
// A is either a dense row-major matrix (optionally split over two sources along K, which is how
// torch.cat([h, skip], 1) feeds a 1x1 shortcut conv without materialising the concat), or the
// im2col view of an NHWC activation (3x3 taps, pad 1; stride 1 / stride 2 / fused nearest-x2
// upsample / stride-2 transposed for dgrad) gathered on the fly — no im2col buffer ever exists.
//
// Layout: activations are NHWC bf16, i.e. a (B*H*W, C) row-major matrix, everywhere in this
// library.  Conv weights are pre-laid-out as [Cout][ky][kx][Cin] (K = 9*Cin contiguous).
//
// Kernel: 4 waves (256 threads) per workgroup, block tile BM x BN x 64, register-staged
// global->LDS copy with the next tile's global loads in flight under the current tile's MFMAs,
// LDS rows padded to 72 bf16 (144 B) so that ds_read_b128 fragment reads are bank-conflict free
// (16 distinct 16-B slots per 16-lane group), v_mfma_f32_32x32x16_bf16 with fp32 accumulation.
// Fused epilogue: alpha, per-column bias, per-(batch,column) bias (the ResBlock time-embedding
// add), residual add (bf16 or fp32), exact-erf GELU, accumulate-into-C, bf16 or fp32 store.
// Optional split-K (grid.z) through an fp32 workspace + a deterministic reduce/epilogue kernel
// (used for the K=11520..23040, M=1024 convs at the 8x8 level and for weight-gradient GEMMs whose
// reduction runs over B*H*W).
//
// Reference call sites this replaces: every F.linear / nn.Linear / nn.Conv2d on the hot path —
// e4t/models/cross_attention.py:506,516,518,534 ; e4t/models/attention.py:376,419-430 ;
// e4t/models/transformer_2d.py:153,205,258-261 ; [3P diffusers] ResnetBlock2D/Downsample2D/Upsample2D
// constructed at e4t/models/unet_2d_blocks.py:481,760,804,881,1732,1774,1855,1872 ;
// e4t/models/unet_2d_condition.py:106-108,285-287 ; [3P open_clip] ViT linears (e4t/encoder.py:154).
#include "gemm_common.h"

#ifndef E4T_GEMM_PS_DEFAULT
#define E4T_GEMM_PS_DEFAULT 0      // automatic choice of the persistent 256 x BN kernel (gemm_ps.hip); E4T_GEMM_PS=0/1 overrides
#endif
extern "C" __attribute__((visibility("hidden"))) int e4t_launch_gemm_ps(const GemmArgs* p, int conv, int bn, int general, int ncu, hipStream_t st);

namespace {


// MODE 0: dense A.  MODE 1: implicit 3x3 conv over NHWC A.
template <int BM, int BN, int WGM, int WGN, int MODE>
__global__ __launch_bounds__(256) void gemm_kernel(GemmArgs p) {
  constexpr int WM = BM / WGM, WN = BN / WGN;  // per-wave tile
  constexpr int FM = WM / 32, FN = WN / 32;    // 32x32 MFMA fragments per wave
  constexpr int NA = BM * (BK / 8) / 256;      // 16-B chunks of A per thread per K-tile
  constexpr int NB = BN * (BK / 8) / 256;
  static_assert(WGM * WGN == 4, "4 waves per workgroup");
  static_assert(NA >= 1 && NB >= 1, "tile too small");

  __shared__ __attribute__((aligned(16))) bf16_t smem[(BM + BN) * LDS_LD];
  bf16_t* const As = smem;
  bf16_t* const Bs = smem + BM * LDS_LD;

  const int tid = threadIdx.x;
  const int lane = tid & 63, wave = tid >> 6;
  const int wm = wave / WGN, wn = wave % WGN;
  const int m0 = blockIdx.y * BM, n0 = blockIdx.x * BN;

  const int nkt = (p.K + BK - 1) / BK;
  const int bz = blockIdx.z / p.splitk, sz = blockIdx.z - bz * p.splitk;
  p.A += bz * p.strideA;
  if (p.A2) p.A2 += bz * p.strideA;
  p.B += bz * p.strideB;
  if (p.bias) p.bias += bz * p.strideBias;
  if (!p.reduce_batch) {
    if (p.flags & E4T_OUT_F32) p.C = (float*)p.C + bz * p.strideC;
    else p.C = (bf16_t*)p.C + bz * p.strideC;
  }
  const int kt_begin = sz * p.ktiles_per_split;
  int kt_end = kt_begin + p.ktiles_per_split;
  if (kt_end > nkt) kt_end = nkt;

  // ---- per-thread chunk ownership (fixed rows across the K loop) ----
  // chunk c = tid + i*256 : row = c >> 3, kc = c & 7  (8 lanes cover one 128-B row segment)
  int a_row[NA];
  long long a_base[NA];  // dense: row offset (elements) ; conv: packed (b, oy, ox)
  int a_oy[NA], a_ox[NA];
  bool a_ok[NA];
#pragma unroll
  for (int i = 0; i < NA; ++i) {
    const int c = tid + i * 256;
    const int r = c >> 3;
    a_row[i] = r;
    const int gr = m0 + r;
    a_ok[i] = gr < p.M;
    if (MODE == 0) {
      a_base[i] = (long long)gr;
      a_oy[i] = a_ox[i] = 0;
    } else {
      const int hw = p.Hout * p.Wout;
      const int b = gr / hw;
      const int rem = gr - b * hw;
      a_oy[i] = rem / p.Wout;
      a_ox[i] = rem - a_oy[i] * p.Wout;
      a_base[i] = (long long)b * p.Hin * p.Win;
    }
  }
  const int kc8 = (tid & 7) * 8;  // this thread's k offset inside a tile (same for all its chunks)

  uint4 ra[NA], rb[NB];

  auto load_tile = [&](int kt) {
    const int k0 = kt * BK;
    const int k = k0 + kc8;
    if (MODE == 0) {
      const bf16_t* src = p.A;
      int ld = p.lda, kk = k;
      if (k0 >= p.K1) { src = p.A2; ld = p.lda2; kk = k - p.K1; }
#pragma unroll
      for (int i = 0; i < NA; ++i) {
        uint4 v = make_uint4(0, 0, 0, 0);
        if (a_ok[i] && k < p.K) v = *(const uint4*)(src + a_base[i] * ld + kk);
        ra[i] = v;
      }
    } else {
      const int tap = k0 / p.Cin;
      const int ci = k - tap * p.Cin;
      const int ky = tap / 3, kx = tap - ky * 3;
#pragma unroll
      for (int i = 0; i < NA; ++i) {
        int iy, ix;
        bool ok = a_ok[i];
        if (p.mode == E4T_CONV_S1) {
          iy = a_oy[i] + ky - 1; ix = a_ox[i] + kx - 1;
          ok = ok && iy >= 0 && iy < p.Hin && ix >= 0 && ix < p.Win;
        } else if (p.mode == E4T_CONV_S2) {
          iy = 2 * a_oy[i] + ky - 1; ix = 2 * a_ox[i] + kx - 1;
          ok = ok && iy >= 0 && iy < p.Hin && ix >= 0 && ix < p.Win;
        } else if (p.mode == E4T_CONV_UP2) {  // logical input = nearest-x2 upsample of X
          iy = a_oy[i] + ky - 1; ix = a_ox[i] + kx - 1;
          ok = ok && iy >= 0 && iy < 2 * p.Hin && ix >= 0 && ix < 2 * p.Win;
          iy >>= 1; ix >>= 1;
        } else if (p.mode == E4T_CONV_S2A) {  // stride 2, pad (0,1,0,1): the VAE encoder's Downsample2D(padding=0)
          iy = 2 * a_oy[i] + ky; ix = 2 * a_ox[i] + kx;
          ok = ok && iy < p.Hin && ix < p.Win;
        } else {  // E4T_CONV_S2T: transposed stride-2 (dgrad of S2); X is the (smaller) output-grad map
          const int sy = a_oy[i] + ky - 1, sx = a_ox[i] + kx - 1;
          ok = ok && sy >= 0 && sx >= 0 && !(sy & 1) && !(sx & 1);
          iy = sy >> 1; ix = sx >> 1;
          ok = ok && iy < p.Hin && ix < p.Win;
        }
        uint4 v = make_uint4(0, 0, 0, 0);
        if (ok) v = *(const uint4*)(p.A + (a_base[i] + (long long)iy * p.Win + ix) * p.Cin + ci);
        ra[i] = v;
      }
    }
#pragma unroll
    for (int i = 0; i < NB; ++i) {
      const int r = (tid + i * 256) >> 3;
      const int gn = n0 + r;
      uint4 v = make_uint4(0, 0, 0, 0);
      if (gn < p.N && k < p.K) v = *(const uint4*)(p.B + (size_t)gn * p.ldb + k);
      rb[i] = v;
    }
  };

  f32x16 acc[FM][FN];
#pragma unroll
  for (int i = 0; i < FM; ++i)
#pragma unroll
    for (int j = 0; j < FN; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

  if (kt_begin < kt_end) load_tile(kt_begin);

  const int frow = lane & 31, fhi = lane >> 5;
  for (int kt = kt_begin; kt < kt_end; ++kt) {
#pragma unroll
    for (int i = 0; i < NA; ++i) *(uint4*)(As + a_row[i] * LDS_LD + kc8) = ra[i];
#pragma unroll
    for (int i = 0; i < NB; ++i) *(uint4*)(Bs + ((tid + i * 256) >> 3) * LDS_LD + kc8) = rb[i];
    __syncthreads();
    if (kt + 1 < kt_end) load_tile(kt + 1);
#pragma unroll
    for (int ks = 0; ks < BK / 16; ++ks) {
      bf16x8 af[FM], bfr[FN];
#pragma unroll
      for (int i = 0; i < FM; ++i)
        af[i] = *(const bf16x8*)(As + (wm * WM + i * 32 + frow) * LDS_LD + ks * 16 + fhi * 8);
#pragma unroll
      for (int j = 0; j < FN; ++j)
        bfr[j] = *(const bf16x8*)(Bs + (wn * WN + j * 32 + frow) * LDS_LD + ks * 16 + fhi * 8);
#pragma unroll
      for (int i = 0; i < FM; ++i)
#pragma unroll
        for (int j = 0; j < FN; ++j)
          acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[i], bfr[j], acc[i][j], 0, 0, 0);
    }
    __syncthreads();
  }

  write_tile<WM, WN, FM, FN, true>(p, acc, wave_stage<WM, WN>(smem, wave), lane, m0 + wm * WM, n0 + wn * WN);
}

// ------------------------------------------------------------------------------------------------
// LDS-DMA variant (the default): operand tiles go HBM -> LDS with global_load_lds_dwordx4, never
// through VGPRs or ds_write (the register-staged kernel above is LDS-write bound: 32 KB of
// ds_write_b128 per K-tile cost about as many LDS cycles as the tile's MFMAs take).
//   * one wave-instruction moves 8 rows x 128 B = 1 KiB, landing lane-linear in LDS, so tiles are
//     UNPADDED [rows][64] bf16; bank conflicts on the ds_read_b128 fragment reads are removed by an XOR
//     swizzle applied on the SOURCE side: LDS chunk slot p of row r holds logical 16-B chunk
//     p ^ ((r >> 1) & 7); fragment reads apply the same involution (16 rows distinct mod 16 -> 16
//     distinct 16-B slots per lane group);
//   * two LDS buffers, ONE barrier per K-tile: the DMA of tile t+1 is issued right after the barrier
//     that publishes tile t and flies under tile t's MFMAs;
//   * out-of-range rows / conv padding / K tail: the lane's source pointer is redirected to a 16-byte
//     zero word in global memory (DMA cannot synthesise zeros).
// ------------------------------------------------------------------------------------------------

#ifdef DMA_TRACE    // debug builds only (tools/dma_trace.sh): cycle stamps of wave 0 of every 8th workgroup
__device__ unsigned long long g_dma_trace[128 * 16];
#define DT(slot) do { if (dt_on) g_dma_trace[dt_wg * 16 + (slot)] = __builtin_readcyclecounter(); } while (0)
#else
#define DT(slot) do { } while (0)
#endif
// KT: K-tile width.  64 (128-byte rows, 8 rows per 1-KiB DMA piece) or 32 (64-byte rows, 16 rows per piece): the same LDS budget
// then holds twice the stages — 4 x 16 KiB instead of 2 x 32 KiB for the 128 x 128 tile, still two workgroups per CU — i.e. 3 K-tiles
// (48 KiB) instead of 1 (32 KiB) in flight per workgroup.  The cycle-stamp trace (tools/dma_trace.sh) shows the K loop of every
// shape waiting ~1700-2500 cycles per 64-wide tile for a DMA issued one iteration earlier against ~500 cycles of MFMA work.
template <int BM, int BN, int WGM, int WGN, int MODE, int NSTAGE, bool GENERAL = false, int KT = 64>
// (second argument = waves per SIMD the register allocation must leave room for: the 2-stage 128 x 160 tile runs TWO 4-wave workgroups per CU — at 260
// registers instead of 256 it ran one, 40 % slower: M16384 N640 K640 22.8 -> 31.9 us, round 6)
__global__ __launch_bounds__(WGM * WGN * 64, (BM == 256 && KT == 32) ? 4 : (BN == 160 ? 2 : 1)) void gemm_dma_kernel(GemmArgs p) {
#ifdef DMA_TRACE
  const int dt_lin = blockIdx.y * gridDim.x + blockIdx.x;
  const int dt_wg = dt_lin >> 3;
  const bool dt_on = (threadIdx.x == 0) && (dt_lin & 7) == 0 && dt_wg < 128 && blockIdx.z == 0;
  DT(0);
#endif
  constexpr int WM = BM / WGM, WN = BN / WGN;
  constexpr int FM = WM / 32, FN = WN / 32;
  constexpr int NW = WGM * WGN;                       // waves per workgroup (4 or 8)
  constexpr int SL = KT / 8;                          // 16-byte slots per LDS row
  constexpr int RPP = 512 / KT;                       // rows per 1-KiB DMA piece (8 or 16)
  constexpr int NA = BM / RPP / NW, NB = (BN / RPP + NW - 1) / NW;   // pieces per wave per K-tile (B: last wave may own fewer)
  constexpr int LOOK = NSTAGE - 1;                    // K-tiles in flight
  constexpr int TILE = (BM + BN) * KT;          // elements per LDS buffer
  static_assert(NA >= 1 && NB >= 1 && (NW == 4 || NW == 8) && (NSTAGE >= 2 && NSTAGE <= 4), "bad tile configuration");
  constexpr bool RAGGED_B = (BN / RPP) % NW != 0;     // the last wave(s) own fewer B pieces: their counted waits use their own count
  static_assert(!RAGGED_B || NB <= 3, "counted vmcnt: per-wave piece counts are enumerated up to 3 B pieces");
  static_assert((KT == 64 || KT == 32) && BM % (RPP * NW) == 0, "bad K-tile width");

  __shared__ __attribute__((aligned(16))) bf16_t smem[NSTAGE * TILE];

  const int tid = threadIdx.x;
  const int lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);      // (an SGPR: the DMA destinations are wave-uniform)
  const int wm = wave / WGN, wn = wave % WGN;
  // B pieces this wave issues per K-tile (wave-uniform): NB, or fewer in the last wave(s) of a ragged split
  const int nbw = RAGGED_B ? min(NB, max(0, BN / RPP - wave * NB)) : NB;
  int tile_x, tile_y;
  xcd_tile(tile_x, tile_y, p.group_m);
  const int m0 = tile_y * BM, n0 = tile_x * BN;

  const int nkt = (p.K + KT - 1) / KT;
  const int bz = blockIdx.z / p.splitk, sz = blockIdx.z - bz * p.splitk;
  p.A += bz * p.strideA;
  if (p.A2) p.A2 += bz * p.strideA;
  p.B += bz * p.strideB;
  if (p.bias) p.bias += bz * p.strideBias;
  if (!p.reduce_batch) {
    if (p.flags & E4T_OUT_F32) p.C = (float*)p.C + bz * p.strideC;
    else p.C = (bf16_t*)p.C + bz * p.strideC;
  }
  const int kt_begin = sz * p.ktiles_per_split * (BK / KT);          // the launcher counts 64-wide tiles
  int kt_end = kt_begin + p.ktiles_per_split * (BK / KT);
  if (kt_end > nkt) kt_end = nkt;

  // Operands are addressed through buffer resources (buffer_load ... lds): a 32-bit per-lane byte offset that changes only
  // when the tile starts a new region — the first tile, a new 3x3 tap (conv), the switch to the second concat source
  // (dense), the ragged last tile — plus a wave-uniform SGPR offset that walks K (+128 B per K-tile).  Issuing a tile costs
  // no VALU at all (a per-tile 64-bit address recomputation cost ~1.2k issue cycles per wave, carried 64-bit pointers still
  // 3 VALU each), and out-of-range rows / conv padding / K tails carry an out-of-range offset: the hardware returns zeros.
  const bool cm = MODE != 0 && p.chan_major;          // channel-chunk-major K order (gemm_common.h, cm_step)
  const __amdgpu_buffer_rsrc_t rs_a = cm ? cm_rsrc(p) : __builtin_amdgcn_make_buffer_rsrc((void*)p.A, 0, (int)p.a_bytes, 0x00020000);
  const __amdgpu_buffer_rsrc_t rs_a2 = __builtin_amdgcn_make_buffer_rsrc((void*)(p.A2 ? p.A2 : p.A), 0, (int)(p.A2 ? p.a2_bytes : p.a_bytes), 0x00020000);
  const __amdgpu_buffer_rsrc_t rs_b = __builtin_amdgcn_make_buffer_rsrc((void*)p.B, 0, (int)p.b_bytes, 0x00020000);
  constexpr unsigned OOB = 0xFFFF0000u;          // >= every extent the launcher accepts
  const int lrow = lane / SL, lslot = lane % SL;   // position of this lane inside a 1-KiB piece
  auto swz = [](int r) { return KT == 64 ? ((r >> 1) & 7) : ((r >> 2) & 3); };      // source-side XOR swizzle of the 16-byte slots

  // rows this lane feeds: piece q = wave*NA + i covers tile rows q*8 .. q*8+7
  long long a_base[NA];
  int a_oy[NA], a_ox[NA], a_kc[NA];
  bool a_ok[NA];
#pragma unroll
  for (int i = 0; i < NA; ++i) {
    const int r = (wave * NA + i) * RPP + lrow;
    const int gr = m0 + r;
    a_ok[i] = gr < p.M;
    a_kc[i] = (lslot ^ swz(r)) * 8;       // logical k offset (elements) this lane fetches for that row
    if (MODE == 0) {
      a_base[i] = (long long)gr; a_oy[i] = a_ox[i] = 0;
    } else {
      const int hw = p.Hout * p.Wout;
      const int b = gr / hw;
      const int rem = gr - b * hw;
      a_oy[i] = rem / p.Wout;
      a_ox[i] = rem - a_oy[i] * p.Wout;
      a_base[i] = (long long)b * p.Hin * p.Win;
    }
  }
  unsigned b_row[NB];
  int b_kc[NB];
  bool b_ok[NB];
#pragma unroll
  for (int i = 0; i < NB; ++i) {
    const int r = (wave * NB + i) * RPP + lrow;
    const int gn = n0 + r;
    b_ok[i] = gn < p.N && r < BN;
    b_kc[i] = (lslot ^ swz(r)) * 8;
    b_row[i] = (unsigned)(((size_t)(b_ok[i] ? gn : 0) * p.ldb + b_kc[i]) * 2);
  }

  unsigned a_vo[NA], b_vo[NB];
  int a_so = 0, b_so = 0;          // wave-uniform byte offsets along K
  bool a_second = false;           // reading the second concat source
  auto place_a = [&](int k0) {
    if (MODE == 0) {
      int ld = p.lda, koff = k0;
      a_second = k0 >= p.K1;
      if (a_second) { ld = p.lda2; koff = k0 - p.K1; }
      a_so = __builtin_amdgcn_readfirstlane(koff * 2);          // (wave-uniform by construction; keeps the offset in an SGPR for the compiler)
#pragma unroll
      for (int i = 0; i < NA; ++i) {
        const bool ok = a_ok[i] && (k0 + a_kc[i] < p.K);
        a_vo[i] = ok ? (unsigned)((a_base[i] * ld + a_kc[i]) * 2) : OOB;
      }
    } else {
      const int tap = k0 / p.Cin;
      const int ci0 = k0 - tap * p.Cin;
      const int ky = tap / 3, kx = tap - ky * 3;
      a_so = __builtin_amdgcn_readfirstlane(ci0 * 2);
#pragma unroll
      for (int i = 0; i < NA; ++i) {
        int iy, ix;
        bool ok = a_ok[i];
        if (p.mode == E4T_CONV_S1) {
          iy = a_oy[i] + ky - 1; ix = a_ox[i] + kx - 1;
          ok = ok && iy >= 0 && iy < p.Hin && ix >= 0 && ix < p.Win;
        } else if (p.mode == E4T_CONV_S2) {
          iy = 2 * a_oy[i] + ky - 1; ix = 2 * a_ox[i] + kx - 1;
          ok = ok && iy >= 0 && iy < p.Hin && ix >= 0 && ix < p.Win;
        } else if (p.mode == E4T_CONV_UP2) {
          iy = a_oy[i] + ky - 1; ix = a_ox[i] + kx - 1;
          ok = ok && iy >= 0 && iy < 2 * p.Hin && ix >= 0 && ix < 2 * p.Win;
          iy >>= 1; ix >>= 1;
        } else if (p.mode == E4T_CONV_S2A) {
          iy = 2 * a_oy[i] + ky; ix = 2 * a_ox[i] + kx;
          ok = ok && iy < p.Hin && ix < p.Win;
        } else {
          const int sy = a_oy[i] + ky - 1, sx = a_ox[i] + kx - 1;
          ok = ok && sy >= 0 && sx >= 0 && !(sy & 1) && !(sx & 1);
          iy = sy >> 1; ix = sx >> 1;
          ok = ok && iy < p.Hin && ix < p.Win;
        }
        a_vo[i] = ok ? (unsigned)(((a_base[i] + (long long)iy * p.Win + ix) * p.Cin + a_kc[i]) * 2) : OOB;
      }
    }
  };
  auto place_b = [&](int k0) {
    b_so = k0 * 2;
#pragma unroll
    for (int i = 0; i < NB; ++i) {
      const bool ok = b_ok[i] && (k0 + b_kc[i] < p.K);
      b_vo[i] = ok ? b_row[i] : OOB;
    }
  };
  if (cm) {                        // a_vo = the output position's own pixel (all taps), a_oy = inverted 9-bit tap validity mask
#pragma unroll
    for (int i = 0; i < NA; ++i) {
      a_vo[i] = cm_center(p, a_base[i], a_oy[i], a_ox[i], a_kc[i]);
      a_oy[i] = cm_inv_mask(p, a_ok[i], a_oy[i], a_ox[i]);
    }
  }
  CmWalk wk;                       // cm: (tap, chunk) of the next K-tile to issue
  wk.init(kt_begin, KT);
  const int cm_table = cm ? cm_tap_table(p, lane) : 0;
  auto issue_tile = [&](int kt, bf16_t* buf) __attribute__((always_inline)) {
    const int k0 = kt * KT;
    bf16_t* As = buf;
    bf16_t* Bs = buf + BM * KT;
    if (cm) {                      // whole K-tiles only (Cin % KT == 0): no ragged tile; tiles are issued in increasing kt
      if (kt == kt_begin) place_b(k0);
      const int aso = cm_a_so(cm_table, wk), bso = cm_b_so(p, wk);
#pragma unroll
      for (int i = 0; i < NA; ++i)
        buf_dma16(rs_a, cm_row_off(a_vo[i], a_oy[i], wk), aso, As + (wave * NA + i) * 512);
#pragma unroll
      for (int i = 0; i < NB; ++i)
        if ((BN / RPP) % NW == 0 || wave * NB + i < BN / RPP)
          buf_dma16(rs_b, b_vo[i], bso, Bs + (wave * NB + i) * 512);
      wk.next(KT);
      return;
    }
    const bool ragged = k0 + KT > p.K;                                      // wave-uniform conditions
    const bool fresh_a = kt == kt_begin || ragged || (MODE == 0 ? k0 == p.K1 : (k0 % p.Cin) == 0);
    if (fresh_a) place_a(k0);
    else a_so += KT * 2;
    if (kt == kt_begin || ragged) place_b(k0);
    else b_so += KT * 2;
#pragma unroll
    for (int i = 0; i < NA; ++i)
      buf_dma16(a_second ? rs_a2 : rs_a, a_vo[i], a_so, As + (wave * NA + i) * 512);
#pragma unroll
    for (int i = 0; i < NB; ++i)
      if ((BN / RPP) % NW == 0 || wave * NB + i < BN / RPP)
        buf_dma16(rs_b, b_vo[i], b_so, Bs + (wave * NB + i) * 512);
  };

  f32x16 acc[FM][FN];
#pragma unroll
  for (int i = 0; i < FM; ++i)
#pragma unroll
    for (int j = 0; j < FN; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

  const int frow = lane & 31, fhi = lane >> 5;
  // fragment offsets inside a stage (elements), one per (fragment, k-step): computed once; the stage base is a
  // compile-time constant of the unrolled loop below, so every ds_read_b128 is "vgpr + immediate"
  int a_off[FM][KT / 16], b_off[FN][KT / 16];
#pragma unroll
  for (int ks = 0; ks < KT / 16; ++ks) {
#pragma unroll
    for (int i = 0; i < FM; ++i) {
      const int r = wm * WM + i * 32 + frow;
      a_off[i][ks] = r * KT + (((ks * 2 + fhi) ^ swz(r)) * 8);
    }
#pragma unroll
    for (int j = 0; j < FN; ++j) {
      const int r = wn * WN + j * 32 + frow;
      b_off[j][ks] = BM * KT + r * KT + (((ks * 2 + fhi) ^ swz(r)) * 8);
    }
  }
  // prologue: LOOK tiles in flight
#pragma unroll
  for (int s = 0; s < LOOK; ++s)
    if (kt_begin + s < kt_end) issue_tile(kt_begin + s, smem + s * TILE);
  DT(1);

  auto body = [&](auto CURc, int kt) {
    constexpr int CUR = decltype(CURc)::value;
    constexpr int NXT = (CUR + LOOK) % NSTAGE;
    // this wave's pieces of tile kt have landed once at most the pieces of the newer tiles in flight (LOOK-1 of them, fewer
    // at the end of the K range) are outstanding: a counted wait, the loads of the deeper stages keep flying
    auto wait_tiles = [&](auto Tc) {          // at most T newer K-tiles of this wave's pieces outstanding
      constexpr int T = decltype(Tc)::value;
      if (!RAGGED_B || nbw == NB) wait_vmcnt<T * (NA + NB)>();
      else if (nbw == NB - 1) wait_vmcnt<T * (NA + (NB > 1 ? NB - 1 : 0))>();
      else if (nbw == NB - 2) wait_vmcnt<T * (NA + (NB > 2 ? NB - 2 : 0))>();
      else wait_vmcnt<T * NA>();
    };
    if (LOOK >= 3 && kt + 2 < kt_end) wait_tiles(std::integral_constant<int, 2>{});
    else if (LOOK >= 2 && kt + 1 < kt_end) wait_tiles(std::integral_constant<int, 1>{});
    else wait_vmcnt<0>();
    loop_barrier();                                    // ... everyone's have; and everyone finished reading slot NXT
#ifdef DMA_TRACE
    if (kt - kt_begin < 8) DT(2 + (kt - kt_begin));
#endif
    const bf16_t* st = smem + CUR * TILE;
    bf16x8 af[2][FM], bfr[2][FN];
#pragma unroll
    for (int i = 0; i < FM; ++i) af[0][i] = *(const bf16x8*)(st + a_off[i][0]);
#pragma unroll
    for (int j = 0; j < FN; ++j) bfr[0][j] = *(const bf16x8*)(st + b_off[j][0]);
    if (kt + LOOK < kt_end) issue_tile(kt + LOOK, smem + NXT * TILE);   // after the first fragment reads are in flight
#pragma unroll
    for (int ks = 0; ks < KT / 16; ++ks) {
      const int c = ks & 1, n = c ^ 1;
      if (ks + 1 < KT / 16) {
#pragma unroll
        for (int i = 0; i < FM; ++i) af[n][i] = *(const bf16x8*)(st + a_off[i][ks + 1]);
#pragma unroll
        for (int j = 0; j < FN; ++j) bfr[n][j] = *(const bf16x8*)(st + b_off[j][ks + 1]);
      }
#pragma unroll
      for (int i = 0; i < FM; ++i)
#pragma unroll
        for (int j = 0; j < FN; ++j)
          acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[c][i], bfr[c][j], acc[i][j], 0, 0, 0);
    }
  };
  {
    int kt = kt_begin;
    for (; kt + NSTAGE <= kt_end; kt += NSTAGE) {
      body(std::integral_constant<int, 0>{}, kt);
      body(std::integral_constant<int, 1>{}, kt + 1);
      if (NSTAGE >= 3) body(std::integral_constant<int, 2 % NSTAGE>{}, kt + 2);
      if (NSTAGE >= 4) body(std::integral_constant<int, 3 % NSTAGE>{}, kt + 3);
    }
    if (kt < kt_end) { body(std::integral_constant<int, 0>{}, kt); ++kt; }
    if (kt < kt_end) { body(std::integral_constant<int, 1>{}, kt); ++kt; }
    if (NSTAGE >= 4 && kt < kt_end) { body(std::integral_constant<int, 2 % NSTAGE>{}, kt); ++kt; }
  }
  DT(10);
  __syncthreads();   // all fragment reads done before the epilogue reuses the LDS
  DT(11);
#ifdef DMA_TRACE
  write_tile<WM, WN, FM, FN, GENERAL>(p, acc, wave_stage<WM, WN>(smem, wave), lane, m0 + wm * WM, n0 + wn * WN, dt_on ? g_dma_trace + dt_wg * 16 : nullptr);
#else
  if constexpr (NW * WM * (WN + 8) > NSTAGE * TILE || WM >= 128) {
    // tall wave tiles (the 4-wave 256-row variants): two row halves, so that the staging fits the operand buffers and the
    // (fully unrolled) epilogue stays at the size of the other kernels'
    static_assert(FM % 2 == 0 && NW * (WM / 2) * (WN + 8) <= NSTAGE * TILE, "epilogue staging must fit");
    write_tile<WM / 2, WN, FM / 2, FN, GENERAL>(p, *(f32x16(*)[FM / 2][FN])(acc + 0), wave_stage<WM / 2, WN>(smem, wave), lane, m0 + wm * WM, n0 + wn * WN);
    __syncthreads();
    write_tile<WM / 2, WN, FM / 2, FN, GENERAL>(p, *(f32x16(*)[FM / 2][FN])(acc + FM / 2), wave_stage<WM / 2, WN>(smem, wave), lane, m0 + wm * WM + WM / 2, n0 + wn * WN);
  } else {
    write_tile<WM, WN, FM, FN, GENERAL>(p, acc, wave_stage<WM, WN>(smem, wave), lane, m0 + wm * WM, n0 + wn * WN);
  }
  // tail rows: only in the 128 x 160 instantiations (and the ping-pong kernels), which have the registers for its 16 loads in flight — inlined
  // into the 64 / 128 / 256 x 128 tiles it cost them an occupancy step (92 -> 162 VGPRs on the 128 x 128 tile); the planner knows (plan_gemm_tail)
  if constexpr (MODE == 0 && BN == 160) {
    static_assert(NSTAGE * TILE * 2 >= NW * 16 * 64 * 4, "tail reduction must fit the operand buffers");
    if (p.tail_rows) gemm_tail<NW, GENERAL>(p, (float*)smem, wave, lane);      // (its first barrier orders it behind the staging reads above)
  }
#endif
  DT(12);
}

// ------------------------------------------------------------------------------------------------
// 3x3 stride-1 convolution on 256-pixel ROW SEGMENTS, the A operand staged as STRIPS (round 6; the VAE's 128-channel level).
//
// gemm_dma_kernel<256, 128, ..., 32> fetches a 256-pixel x 32-channel A tile per (tap, channel chunk): the three taps of one kernel
// row read the SAME pixels shifted by one, three times through the CU's fill path (a timing probe that let two of three taps fetch
// nothing ran the 128 -> 128 conv at 512 x 512 in 1223 instead of 1430 us).  When a tile is 256 consecutive pixels of ONE image row
// (W % 256 == 0), the pixels x0 - 1 .. x0 + 256 of input row y + ky - 1 are staged ONCE per (ky, chunk) — a strip of 258 rows x 64
// bytes — and the taps kx = 0, 1, 2 read it at row offsets 0, 1, 2: the fragment read of output pixel r for tap kx is strip row
// r + kx (any 16 consecutive rows of the (r >> 2) & 3 slot swizzle are conflict-free, so the shift costs nothing in LDS).  A-side fill
// bytes and DMA instructions per MFMA drop 3 x / 2 x; the B side (weights, one 128 x 32 tile per tap and chunk) is unchanged.
//   LDS: two strip buffers of 24 pieces (384 rows; rows beyond 257 and pixels outside the image carry the out-of-range offset and
//   fetch nothing) + three B buffers = 72 KB, two workgroups per CU, the epilogue staging of gemm_dma_kernel on top of it.
//   Order of the K walk: ky, channel chunk, kx (B follows: column (3 ky + kx) Cin + 32 chunk): only the fp32 summation order differs.
//   Pipeline: sub-step j = (strip s, kx) issues B(j + 2) and, at kx = 0, strip s + 1; per wave 3 strip pieces + 1 B piece, so the
//   counted waits are vmcnt(1) / (4) / (1) for kx = 0 / 1 / 2 (what may still be in flight behind the B tile — and strip — the sub-step
//   needs), one barrier per sub-step as in the tile kernel.
// ------------------------------------------------------------------------------------------------
#ifndef CONV_STRIP
#define CONV_STRIP 1
#endif
// WGN = 2: 256 x 128 tile, 8 waves, two workgroups per CU.  WGN = 4: 256 x 256 tile, 16 waves in ONE workgroup per CU — the same four waves
// per SIMD, a strip now feeds 256 output channels (half the A-side fill per flop again), and the waves split the issue work: waves 0-7
// stage the strips (3 pieces each per strip), waves 8-15 the 256 x 32 B tiles (2 pieces each per sub-step), each group with its own
// counted wait (vmcnt(0) once per strip / vmcnt(2) per sub-step).
template <int WGN>
__global__ __launch_bounds__(256 * WGN, 4) void conv_strip_kernel(GemmArgs p) {
  constexpr int BN = 64 * WGN, KT = 32, WM = 64, WN = 64, FM = 2, FN = 2;
  constexpr bool SPLIT = WGN == 4;                                   // issue roles split between wave groups
  constexpr int NBP = SPLIT ? 2 : 1;                                 // B pieces per issuing wave and sub-step
  constexpr int ASZ = 384 * KT, BSZ = BN * KT;                       // elements per strip / B buffer
  __shared__ __attribute__((aligned(16))) bf16_t smem[2 * ASZ + 3 * BSZ];
  const int tid = threadIdx.x;
  const int lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wave / WGN, wn = wave % WGN;
  int tile_x, tile_y;
  xcd_tile(tile_x, tile_y, p.group_m);
  const int m0 = tile_y * 256, n0 = tile_x * BN;
  const int hw = p.Hin * p.Win;
  const int b = m0 / hw, rem = m0 - b * hw, y = rem / p.Win, x0 = rem - y * p.Win;
  // the tile's 256 output pixels: one 256-pixel segment of an image row (W % 256 == 0), or 256 / W whole rows (W = 16 .. 128: x0 = 0).
  // A strip = those rows of input row y + ky - 1 (+ ry), each widened by one pixel on both sides: SW = min(W, 256) + 2 strip rows per image row
  const int segw = p.Win < 256 ? p.Win : 256, SW = segw + 2, nrows = 256 / segw;
  const int ncc = p.Cin / KT, nstrip = 3 * ncc, nsub = 3 * nstrip;
  constexpr unsigned OOB = 0xFFFF0000u;
  auto swz = [](int r) { return (r >> 2) & 3; };

  const __amdgpu_buffer_rsrc_t rs_a = cm_rsrc(p);                   // base lowered by one image row + one pixel (never touched: masked lanes)
  const __amdgpu_buffer_rsrc_t rs_b = __builtin_amdgcn_make_buffer_rsrc((void*)p.B, 0, (int)p.b_bytes, 0x00020000);
  // strip pieces of this wave: piece q = wave + 8 i covers strip rows 16 q .. 16 q + 15; strip row j = input pixel x0 - 1 + j
  unsigned a_vo[3];
  int a_inv[3];                                  // bit ky set = this strip row reads padding (or nothing) for kernel row ky
#pragma unroll
  for (int i = 0; i < 3; ++i) {
    const int j = 16 * (wave + 8 * i) + (lane >> 2);
    const int ry = j / SW, xs = j - ry * SW, ix = x0 - 1 + xs;
    const bool ok = ry < nrows && ix >= 0 && ix < p.Win;
    a_vo[i] = (unsigned)(((ry * p.Win + xs) * p.Cin + ((lane & 3) ^ swz(j)) * 8) * 2);
    int m = 0;
#pragma unroll
    for (int ky = 0; ky < 3; ++ky) {
      const int iy = y + ry + ky - 1;
      if (!(ok && iy >= 0 && iy < p.Hin)) m |= 1 << ky;
    }
    a_inv[i] = m;
  }
  const bool a_wave = !SPLIT || wave < 8, b_wave = !SPLIT || wave >= 8;      // (wave-uniform)
  const int bw = SPLIT ? wave - 8 : wave;                                     // B pieces of this wave: NBP bw .. NBP bw + NBP - 1
  unsigned b_vo[NBP];
#pragma unroll
  for (int i = 0; i < NBP; ++i) {
    const int r = 16 * (NBP * bw + i) + (lane >> 2);
    b_vo[i] = (b_wave && n0 + r < p.N) ? (unsigned)((((size_t)(n0 + r)) * p.ldb + ((lane & 3) ^ swz(r)) * 8) * 2) : OOB;
  }
  // walkers (incremental: no division per issue)
  int a_ky = 0, a_cc = 0;                        // next strip to issue
  int b_ky = 0, b_cc = 0, b_kx = 0;              // next B tile to issue
  auto issue_a = [&](bf16_t* buf) __attribute__((always_inline)) {
    const int iy = y + a_ky - 1;
    // ((b H + iy) W + x0 - 1) Cin + 32 cc, against the lowered base: + (W + 1) Cin  ->  ((b H + iy + 1) W + x0) Cin + 32 cc  >= 0
    const int so = __builtin_amdgcn_readfirstlane((int)(((((long long)b * p.Hin + iy + 1) * p.Win + x0) * p.Cin + a_cc * KT) * 2));
#pragma unroll
    for (int i = 0; i < 3; ++i)      // the lane's offset, or all-ones (out of range) where the strip row is padding for this kernel row
      buf_dma16(rs_a, a_vo[i] | (unsigned)__builtin_amdgcn_sbfe(a_inv[i], (unsigned)a_ky, 1u), so, buf + (wave + 8 * i) * 512);
    if (++a_cc == ncc) { a_cc = 0; ++a_ky; }
  };
  auto issue_b = [&](bf16_t* buf) __attribute__((always_inline)) {
    const int so = __builtin_amdgcn_readfirstlane((((b_ky * 3 + b_kx) * p.Cin) + b_cc * KT) * 2);
#pragma unroll
    for (int i = 0; i < NBP; ++i) buf_dma16(rs_b, b_vo[i], so, buf + (NBP * bw + i) * 512);
    if (++b_kx == 3) { b_kx = 0; if (++b_cc == ncc) { b_cc = 0; ++b_ky; } }
  };

  f32x16 acc[FM][FN];
#pragma unroll
  for (int i = 0; i < FM; ++i)
#pragma unroll
    for (int j = 0; j < FN; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
  const int frow = lane & 31, fhi = lane >> 5;
  int a_off[3][FM][2], b_off[FN][2];
#pragma unroll
  for (int ks = 0; ks < 2; ++ks) {
#pragma unroll
    for (int kx = 0; kx < 3; ++kx)
#pragma unroll
      for (int i = 0; i < FM; ++i) {
        const int o = wm * WM + i * 32 + frow;                 // output pixel of the tile
        const int r = (o / segw) * SW + o % segw + kx;         // its strip row for tap kx
        a_off[kx][i][ks] = r * KT + (((ks * 2 + fhi) ^ swz(r)) * 8);
      }
#pragma unroll
    for (int j = 0; j < FN; ++j) {
      const int r = wn * WN + j * 32 + frow;
      b_off[j][ks] = r * KT + (((ks * 2 + fhi) ^ swz(r)) * 8);
    }
  }
  bf16_t* const Bs = smem + 2 * ASZ;
  if (a_wave) issue_a(smem);                     // strip 0
  if (b_wave) { issue_b(Bs); issue_b(Bs + BSZ); }      // B(0), B(1)

  auto sub = [&](auto par_, auto kx_, int s) __attribute__((always_inline)) {
    constexpr int PAR = decltype(par_)::value, KX = decltype(kx_)::value;
    const int j = 3 * s + KX;
    if (SPLIT) {
      if (a_wave) { if (KX == 0) wait_vmcnt<0>(); }             // strip s (issued three sub-steps ago) is this wave's only traffic
      else if (j + 1 == nsub) wait_vmcnt<0>();                  // the last B tile
      else wait_vmcnt<2>();                                     // B(j + 1) may still fly
    } else {
      if (s + 1 == nstrip && KX > 0) wait_vmcnt<0>();          // the last strip: nothing (or one B tile) behind the one needed
      else if (KX == 1) wait_vmcnt<4>();                        // strip s + 1 (3 pieces) and B(j + 1) may still fly
      else wait_vmcnt<1>();                                     // B(j + 1) may
    }
    loop_barrier();
    if (a_wave && KX == 0 && s + 1 < nstrip) issue_a(smem + (PAR ^ 1) * ASZ);
    if (b_wave && j + 2 < nsub) issue_b(Bs + ((KX + 2) % 3) * BSZ);
    const bf16_t* sa = smem + PAR * ASZ;
    const bf16_t* sb = Bs + KX * BSZ;
    bf16x8 af[2][FM], bfr[2][FN];
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) {
#pragma unroll
      for (int i = 0; i < FM; ++i) af[ks][i] = *(const bf16x8*)(sa + a_off[KX][i][ks]);
#pragma unroll
      for (int jj = 0; jj < FN; ++jj) bfr[ks][jj] = *(const bf16x8*)(sb + b_off[jj][ks]);
    }
#pragma unroll
    for (int ks = 0; ks < 2; ++ks)
#pragma unroll
      for (int i = 0; i < FM; ++i)
#pragma unroll
        for (int jj = 0; jj < FN; ++jj)
          acc[i][jj] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[ks][i], bfr[ks][jj], acc[i][jj], 0, 0, 0);
  };
  using I0 = std::integral_constant<int, 0>;
  using I1 = std::integral_constant<int, 1>;
  using I2 = std::integral_constant<int, 2>;
  for (int s = 0; s < nstrip; s += 2) {
    sub(I0{}, I0{}, s); sub(I0{}, I1{}, s); sub(I0{}, I2{}, s);
    if (s + 1 < nstrip) { sub(I1{}, I0{}, s + 1); sub(I1{}, I1{}, s + 1); sub(I1{}, I2{}, s + 1); }
  }
  __syncthreads();   // all fragment reads done before the epilogue reuses the LDS
  if constexpr (SPLIT) {     // 16 staging areas of a whole wave tile would not fit the operand buffers: two row halves (as the tall gemm_dma tiles do)
    static_assert(4 * WGN * (WM / 2) * (WN + 8) <= 2 * ASZ + 3 * BSZ, "epilogue staging must fit");
    write_tile<WM / 2, WN, FM / 2, FN, false>(p, *(f32x16(*)[FM / 2][FN])(acc + 0), wave_stage<WM / 2, WN>(smem, wave), lane, m0 + wm * WM, n0 + wn * WN);
    __syncthreads();
    write_tile<WM / 2, WN, FM / 2, FN, false>(p, *(f32x16(*)[FM / 2][FN])(acc + FM / 2), wave_stage<WM / 2, WN>(smem, wave), lane, m0 + wm * WM + WM / 2, n0 + wn * WN);
  } else {
    write_tile<WM, WN, FM, FN, false>(p, acc, wave_stage<WM, WN>(smem, wave), lane, m0 + wm * WM, n0 + wn * WN);
  }
}
// may a 256 x 128 x 32 (WGN = 2) / 256 x 256 (WGN = 4, in place of the ping-pong kernel) conv launch go to conv_strip_kernel?
static bool conv_strip_ok(const GemmArgs& p, int splitk, int batch) {
  static const bool on = CONV_STRIP && getenv("E4T_CONV_NOSTRIP") == nullptr;          // A/B switch
  const bool rows = p.Win % 256 == 0 || (p.Win >= 16 && 256 % p.Win == 0 && p.Hin % (256 / p.Win) == 0);      // row segments, or whole rows per tile
  return on && p.mode == E4T_CONV_S1 && p.chan_major && splitk == 1 && batch == 1 && rows && p.Wout == p.Win && p.Hout == p.Hin &&
         p.Cin % 32 == 0 && p.K == 9 * p.Cin;
}

#ifdef DMA_TRACE
}  // namespace
extern "C" int e4t_debug_dma_trace(unsigned long long* out) {
  return (int)hipMemcpyFromSymbol(out, HIP_SYMBOL(g_dma_trace), sizeof(g_dma_trace));
}
namespace {
#endif

// ------------------------------------------------------------------------------------------------
// 256 x 256 x 64 "ping-pong" variant for the big, K-deep shapes (VAE / 1280-channel convs, FF GEMMs).
//
// The 128-wide tiles above need ~64 B/clk/CU of L2->LDS traffic at MFMA peak and hide latency by occupancy
// (2 workgroups x 8 waves); they top out near 1 PFLOP/s.  This one halves the operand traffic per flop and
// schedules explicitly instead:
//   * 8 waves = 2 groups (wr) x 4 column slices (wc), wave tile 128 x 64 (4 x 2 accumulators of 32x32);
//     wave w and w+4 share a SIMD, one from each group;
//   * a K-tile is consumed in 4 PHASES of 8 MFMAs: (rows 0-63 | 64-127 of the wave tile) x (k 0-31 | 32-63).
//     Each phase = [L: fragment ds_reads + one 16-KiB operand quarter DMA issue + counted vmcnt] barrier
//     [M: 8 MFMAs at raised priority] barrier.  Group 1 runs one barrier interval behind group 0, so on every
//     SIMD one wave is in its MFMA segment while the other does its LDS/DMA segment;
//   * the operand stream is cut in QUARTERS (A|B) x (k-lo|k-hi), 256 rows x 32 k = 16 KiB each, two buffers of four:
//     a quarter slot is re-staged exactly 2 phases after its last ds_read and first read >= 5 phases after its
//     issue, so the DMA has > 2500 clk of flight time and vmcnt never drains to 0 inside the loop
//     (each phase issues 2 DMA instructions per wave; vmcnt(8) after the issue = everything older than the 4
//     newest quarters has landed; a quarter is read one phase after the wait + barrier that retires it);
//   * quarter rows are 64 B: LDS slot p of row r holds logical 16-B chunk p ^ ((r >> 2) & 3) (source-side swizzle,
//     conflict-free ds_read_b128 for the 32x32x16 fragment lane groups).
// Requires K % 64 == 0 (no ragged K-tile); M / N edges are handled by zero rows and the bounds-checked epilogue.
// ------------------------------------------------------------------------------------------------
#ifdef PP_TRACE     // debug builds only (tools/pp_trace.sh): cycle stamps of one wave per group
__device__ unsigned long long g_pp_trace[2 * 6 * 64];
#ifdef PP_TRACE_MIN
#define PP_STAMP(slot) do { if ((slot) == 0 && tracing && tr_n < 64) tr[wr][tr_n * 6] = __builtin_readcyclecounter(); } while (0)
#else
#define PP_STAMP(slot) do { if (tracing && tr_n < 64) tr[wr][tr_n * 6 + (slot)] = __builtin_readcyclecounter(); } while (0)
#endif
#else
#define PP_STAMP(slot) do { } while (0)
#endif
// timing ablations of the debug trace build (results are garbage): -DPP_NOREAD skips the fragment reads, -DPP_NOWAIT the vmcnt waits
#ifdef PP_NOREAD
#define PP_LDSREAD(ptr) bf16x8{}
#else
#define PP_LDSREAD(ptr) (*(const bf16x8*)(ptr))
#endif
// CM (round 5): the channel-chunk-major K order of the stride-1 3x3 convs is a compile-time property of the instantiation, and the K loop
// runs its steady state — every quarter issue unconditional — apart from the last tiles.  As a runtime flag (round 3) both walks lived in
// every phase: a branch on the flag per DMA issue, the tap-major offsets kept alive next to the channel-major ones (the wave-uniform
// A offset was parked in a VGPR and fetched back with v_readfirstlane per issue), and the "is there a tile left to stage" compare /
// select / branch chain in every phase: ~9 of the ~24 scalar instructions per 8-MFMA phase (ISA, profiles/r05_isa_pp_loop.txt).
template <int MODE, bool GENERAL = false, bool CM = false>
__global__ __launch_bounds__(512) void gemm_pp_kernel(GemmArgs p) {
  constexpr int BM = 256, BN = 256, HK = 32;
  constexpr int QUART = 256 * HK;                        // elements per quarter
  constexpr int EPI = 8 * 64 * (64 + 8);                 // write_tile staging: 8 waves x 64 x 72
  constexpr int SMEM = EPI > 8 * QUART ? EPI : 8 * QUART;
  __shared__ __attribute__((aligned(16))) bf16_t smem[SMEM];

  const int tid = threadIdx.x;
  const int lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wr = wave >> 2, wc = wave & 3;
#ifdef PP_TRACE
  __shared__ unsigned long long tr[2][6 * 64];
  const bool tracing = blockIdx.x == 1 && blockIdx.y == 1 && blockIdx.z == 0 && wc == 0 && lane == 0;
  int tr_n = -32;       // skip the first 32 phases
#endif
  int tile_x, tile_y;
  xcd_tile(tile_x, tile_y, p.group_m);
  const int m0 = tile_y * BM, n0 = tile_x * BN;

  const int nkt = p.K / BK;
  const int bz = blockIdx.z / p.splitk, sz = blockIdx.z - bz * p.splitk;
  p.A += bz * p.strideA;
  if (p.A2) p.A2 += bz * p.strideA;
  p.B += bz * p.strideB;
  if (p.bias) p.bias += bz * p.strideBias;
  if (!p.reduce_batch) {
    if (p.flags & E4T_OUT_F32) p.C = (float*)p.C + bz * p.strideC;
    else p.C = (bf16_t*)p.C + bz * p.strideC;
  }
  const int kt_begin = sz * p.ktiles_per_split;
  int kt_end = kt_begin + p.ktiles_per_split;
  if (kt_end > nkt) kt_end = nkt;

  // Operands are addressed through buffer resources (buffer_load ... lds): a 32-bit per-lane byte offset that changes only
  // when the conv tap / concat source changes, plus a wave-uniform SGPR offset that walks K (+64 B per quarter).  Issuing a
  // quarter costs no VALU at all, and out-of-range rows / conv padding simply carry an out-of-range offset (the hardware
  // returns zeros) instead of a redirected pointer.
  constexpr bool cm = MODE != 0 && CM;               // channel-chunk-major K order (gemm_common.h, cm_step): launch_gemm picks the instantiation
  const __amdgpu_buffer_rsrc_t rs_a = cm ? cm_rsrc(p) : uniform_rsrc(p.A, p.a_bytes);
  const __amdgpu_buffer_rsrc_t rs_a2 = uniform_rsrc(p.A2 ? p.A2 : p.A, p.A2 ? p.a2_bytes : p.a_bytes);
  const __amdgpu_buffer_rsrc_t rs_b = uniform_rsrc(p.B, p.b_bytes);
  constexpr unsigned OOB = 0xFFFF0000u;          // >= every extent the launcher accepts
  // DMA: one wave-instruction = 16 rows x 64 B; wave w feeds quarter rows 32w + 16j + (lane >> 2), j = 0, 1
  const int drow = lane >> 2, dslot = lane & 3;
  long long a_base[2];
  int a_oy[2], a_ox[2], a_kc[2];
  bool a_ok[2];
  unsigned b_vo[2];
#pragma unroll
  for (int j = 0; j < 2; ++j) {
    const int r = wave * 32 + j * 16 + drow;
    const int kc = (dslot ^ ((r >> 2) & 3)) * 8;
    const int gr = m0 + r;
    a_ok[j] = gr < p.M;
    a_kc[j] = kc;
    if (MODE == 0) {
      a_base[j] = (long long)gr; a_oy[j] = a_ox[j] = 0;
    } else {
      const int hw = p.Hout * p.Wout;
      const int b = gr / hw;
      const int rem = gr - b * hw;
      a_oy[j] = rem / p.Wout;
      a_ox[j] = rem - a_oy[j] * p.Wout;
      a_base[j] = (long long)b * p.Hin * p.Win;
    }
    const int gn = n0 + r;
    b_vo[j] = gn < p.N ? (unsigned)(((size_t)gn * p.ldb + kc) * 2) : OOB;
  }
  unsigned a_vo[2];
  if (cm) {                        // a_vo = the output position's own pixel (all taps), a_oy = inverted 9-bit tap validity mask
#pragma unroll
    for (int j = 0; j < 2; ++j) {
      a_vo[j] = cm_center(p, a_base[j], a_oy[j], a_ox[j], a_kc[j]);
      a_oy[j] = cm_inv_mask(p, a_ok[j], a_oy[j], a_ox[j]);
    }
  }
  unsigned a_eff[2] = {0u, 0u};    // cm: the offsets of the K-tile whose quarters are being issued
  CmWalk wa, wb;                   // cm: one walker per operand stream (A and B are issued at different times)
  wa.init(kt_begin, BK); wb = wa;
  const int cm_table = cm ? cm_tap_table(p, lane) : 0;
  int a_so = 0, b_so = 0;          // wave-uniform byte offsets along K
  bool a_second = false;           // reading the second concat source
  auto place_a = [&](int k0) {
    if (MODE == 0) {
      int ld = p.lda, koff = k0;
      a_second = k0 >= p.K1;
      if (a_second) { ld = p.lda2; koff = k0 - p.K1; }
      a_so = __builtin_amdgcn_readfirstlane(koff * 2);          // (wave-uniform by construction; keeps the offset in an SGPR for the compiler)
#pragma unroll
      for (int j = 0; j < 2; ++j) a_vo[j] = a_ok[j] ? (unsigned)((a_base[j] * ld + a_kc[j]) * 2) : OOB;
    } else {
      const int tap = k0 / p.Cin;
      const int ci0 = k0 - tap * p.Cin;
      const int ky = tap / 3, kx = tap - ky * 3;
      a_so = __builtin_amdgcn_readfirstlane(ci0 * 2);
#pragma unroll
      for (int j = 0; j < 2; ++j) {
        int iy, ix;
        bool ok = a_ok[j];
        if (p.mode == E4T_CONV_S1) {
          iy = a_oy[j] + ky - 1; ix = a_ox[j] + kx - 1;
          ok = ok && iy >= 0 && iy < p.Hin && ix >= 0 && ix < p.Win;
        } else if (p.mode == E4T_CONV_S2) {
          iy = 2 * a_oy[j] + ky - 1; ix = 2 * a_ox[j] + kx - 1;
          ok = ok && iy >= 0 && iy < p.Hin && ix >= 0 && ix < p.Win;
        } else if (p.mode == E4T_CONV_UP2) {
          iy = a_oy[j] + ky - 1; ix = a_ox[j] + kx - 1;
          ok = ok && iy >= 0 && iy < 2 * p.Hin && ix >= 0 && ix < 2 * p.Win;
          iy >>= 1; ix >>= 1;
        } else if (p.mode == E4T_CONV_S2A) {
          iy = 2 * a_oy[j] + ky; ix = 2 * a_ox[j] + kx;
          ok = ok && iy < p.Hin && ix < p.Win;
        } else {
          const int sy = a_oy[j] + ky - 1, sx = a_ox[j] + kx - 1;
          ok = ok && sy >= 0 && sx >= 0 && !(sy & 1) && !(sx & 1);
          iy = sy >> 1; ix = sx >> 1;
          ok = ok && iy < p.Hin && ix < p.Win;
        }
        a_vo[j] = ok ? (unsigned)(((a_base[j] + (long long)iy * p.Win + ix) * p.Cin + a_kc[j]) * 2) : OOB;
      }
    }
  };
  // The A and B streams are each issued in increasing k (lo(t), hi(t), lo(t+1), ...): +64 bytes on the scalar offset per quarter.
  auto issue_a = [&](int kt, bool hi, bf16_t* dst) __attribute__((always_inline)) {
    if (cm) {                        // lo(t), hi(t), lo(t+1), ...: the walker advances after each hi
      if (!hi) {
        a_so = cm_a_so(cm_table, wa);
#pragma unroll
        for (int j = 0; j < 2; ++j) a_eff[j] = cm_row_off(a_vo[j], a_oy[j], wa);
      } else {
        a_so = __builtin_amdgcn_readfirstlane(a_so + HK * 2);
      }
#pragma unroll
      for (int j = 0; j < 2; ++j) buf_dma16(rs_a, a_eff[j], a_so, dst + (wave * 32 + j * 16) * HK);
      if (hi) wa.next(BK);
      return;
    }
    const int k0 = kt * BK;
    const bool fresh = !hi && (kt == kt_begin || (MODE == 0 ? k0 == p.K1 : (k0 % p.Cin) == 0));
    if (fresh) place_a(k0);
    else a_so = __builtin_amdgcn_readfirstlane(a_so + HK * 2);
#pragma unroll
    for (int j = 0; j < 2; ++j)
      buf_dma16(a_second ? rs_a2 : rs_a, a_vo[j], a_so, dst + (wave * 32 + j * 16) * HK);
  };
  auto issue_b = [&](int kt, bool hi, bf16_t* dst) __attribute__((always_inline)) {
    if (cm && !hi) b_so = cm_b_so(p, wb);
    else if (!cm && !hi && kt == kt_begin) b_so = kt * BK * 2;
    else b_so = __builtin_amdgcn_readfirstlane(b_so + HK * 2);
#pragma unroll
    for (int j = 0; j < 2; ++j)
      buf_dma16(rs_b, b_vo[j], b_so, dst + (wave * 32 + j * 16) * HK);
    if (cm && hi) wb.next(BK);
  };

  f32x16 acc[4][2];
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

  const int frow = lane & 31, fhi = lane >> 5;
  int a_off[4][2], b_off[2][2];     // fragment offsets inside a quarter (elements), [block][k-step of the half]
#pragma unroll
  for (int ks = 0; ks < 2; ++ks) {
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int r = wr * 128 + i * 32 + frow;
      a_off[i][ks] = r * HK + (((ks * 2 + fhi) ^ ((r >> 2) & 3)) * 8);
    }
#pragma unroll
    for (int j = 0; j < 2; ++j) {
      const int r = wc * 64 + j * 32 + frow;
      b_off[j][ks] = r * HK + (((ks * 2 + fhi) ^ ((r >> 2) & 3)) * 8);
    }
  }
  // quarter slots of buffer b: lo-A, lo-B, hi-A, hi-B at (4b + 0..3) * QUART
  issue_b(kt_begin, false, smem + 1 * QUART);
  issue_a(kt_begin, false, smem + 0 * QUART);
  issue_b(kt_begin, true, smem + 3 * QUART);
  issue_a(kt_begin, true, smem + 2 * QUART);
  if (kt_begin + 1 < kt_end) {
    issue_b(kt_begin + 1, false, smem + 5 * QUART);
    issue_a(kt_begin + 1, false, smem + 4 * QUART);
    wait_vmcnt<8>();               // 12 issued: the two lo quarters of the first K-tile have landed
  } else {
    wait_vmcnt<0>();
  }
  __builtin_amdgcn_s_barrier();

  bf16x8 af[2][2], bfr[2][2];
  auto phase = [&](auto Bc, auto Pc, auto ALLc, int kt) {
    constexpr int b = decltype(Bc)::value, ph = decltype(Pc)::value;
    constexpr bool ALL = decltype(ALLc)::value;      // steady state: every quarter this phase stages exists
    constexpr int kh = ph >> 1, mh = ph & 1;
    bf16_t* const buf = smem + b * 4 * QUART;
    bf16_t* const other = smem + (b ^ 1) * 4 * QUART;
    const bf16_t* const qa = buf + (kh * 2) * QUART;
    const bf16_t* const qb = qa + QUART;
    // ---- L segment ----
    PP_STAMP(0);
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
      for (int ks = 0; ks < 2; ++ks) af[i][ks] = PP_LDSREAD(qa + a_off[mh * 2 + i][ks]);
    if (mh == 0) {
#pragma unroll
      for (int j = 0; j < 2; ++j)
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) bfr[j][ks] = PP_LDSREAD(qb + b_off[j][ks]);
    }
    bool staged;
    if (ph == 0)      { staged = ALL || kt + 1 < kt_end; if (staged) issue_b(kt + 1, true, other + 3 * QUART); }
    else if (ph == 1) { staged = ALL || kt + 1 < kt_end; if (staged) issue_a(kt + 1, true, other + 2 * QUART); }
    else if (ph == 2) { staged = ALL || kt + 2 < kt_end; if (staged) issue_b(kt + 2, false, buf + 1 * QUART); }
    else              { staged = ALL || kt + 2 < kt_end; if (staged) issue_a(kt + 2, false, buf + 0 * QUART); }
    PP_STAMP(1);
#ifndef PP_NOWAIT
    if (ALL || staged) wait_vmcnt<8>(); else wait_vmcnt<0>();
#endif
    PP_STAMP(2);
    __builtin_amdgcn_sched_barrier(0);
    __builtin_amdgcn_s_barrier();
    __builtin_amdgcn_sched_barrier(0);
    PP_STAMP(3);
    // ---- M segment ----
    __builtin_amdgcn_s_setprio(1);
#pragma unroll
    for (int ks = 0; ks < 2; ++ks)
#pragma unroll
      for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
          acc[mh * 2 + i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[i][ks], bfr[j][ks], acc[mh * 2 + i][j], 0, 0, 0);
    __builtin_amdgcn_s_setprio(0);
    PP_STAMP(4);
    __builtin_amdgcn_sched_barrier(0);
    __builtin_amdgcn_s_barrier();
    __builtin_amdgcn_sched_barrier(0);
    PP_STAMP(5);
#ifdef PP_TRACE
    if (tracing || tr_n < 0) ++tr_n;
#endif
  };
  using I0 = std::integral_constant<int, 0>; using I1 = std::integral_constant<int, 1>;
  using I2 = std::integral_constant<int, 2>; using I3 = std::integral_constant<int, 3>;

  if (wr == 1) __builtin_amdgcn_s_barrier();        // group 1 runs one barrier interval behind group 0
  {
    using YES = std::true_type; using NO = std::false_type;
    int kt = kt_begin;
    for (; kt + 4 <= kt_end; kt += 2) {       // steady state: tiles kt + 2 and kt + 3 exist, nothing to decide
      phase(I0{}, I0{}, YES{}, kt); phase(I0{}, I1{}, YES{}, kt); phase(I0{}, I2{}, YES{}, kt); phase(I0{}, I3{}, YES{}, kt);
      phase(I1{}, I0{}, YES{}, kt + 1); phase(I1{}, I1{}, YES{}, kt + 1); phase(I1{}, I2{}, YES{}, kt + 1); phase(I1{}, I3{}, YES{}, kt + 1);
    }
    for (; kt + 2 <= kt_end; kt += 2) {
      phase(I0{}, I0{}, NO{}, kt); phase(I0{}, I1{}, NO{}, kt); phase(I0{}, I2{}, NO{}, kt); phase(I0{}, I3{}, NO{}, kt);
      phase(I1{}, I0{}, NO{}, kt + 1); phase(I1{}, I1{}, NO{}, kt + 1); phase(I1{}, I2{}, NO{}, kt + 1); phase(I1{}, I3{}, NO{}, kt + 1);
    }
    if (kt < kt_end) { phase(I0{}, I0{}, NO{}, kt); phase(I0{}, I1{}, NO{}, kt); phase(I0{}, I2{}, NO{}, kt); phase(I0{}, I3{}, NO{}, kt); }
  }
  if (wr == 0) __builtin_amdgcn_s_barrier();
  __syncthreads();   // every fragment read and every DMA is done before the epilogue reuses the LDS
#ifdef PP_TRACE
  if (tracing) for (int i = 0; i < 6 * 64; ++i) g_pp_trace[wr * 6 * 64 + i] = tr[wr][i];
#endif
  // two 64-row halves: keeps the (fully unrolled) epilogue at the size of the 128-wide kernels'
  write_tile<64, 64, 2, 2, GENERAL>(p, *(f32x16(*)[2][2])(acc + 0), wave_stage<64, 64>(smem, wave), lane, m0 + wr * 128, n0 + wc * 64);
  __syncthreads();
  write_tile<64, 64, 2, 2, GENERAL>(p, *(f32x16(*)[2][2])(acc + 2), wave_stage<64, 64>(smem, wave), lane, m0 + wr * 128 + 64, n0 + wc * 64);
  if constexpr (MODE == 0) {
    if (p.tail_rows) gemm_tail<8, GENERAL>(p, (float*)smem, wave, lane);
  }
}

// ------------------------------------------------------------------------------------------------
// gemm_pp_kernel for stride-1 3x3 convs with the A operand staged as STRIPS (round 6; see conv_strip_kernel for the idea and the probe
// behind it: a third of the A-side fill is worth 12-16 % to the ping-pong kernel on every conv shape of the step).
//
// Same machine — two groups of four waves alternating an LDS/DMA segment with an 8-MFMA segment, B quarters of 256 x 32 by LDS-DMA, the
// channel-chunk-major K order (K-tile kt = chunk kt / 9, tap kt % 9) — but the three K-tiles kx = 0, 1, 2 of one kernel row read ONE strip:
// the tile's 256 output pixels (a 256-pixel row segment, or 256 / W whole rows) of input row y + ky - 1, each image row widened by a pixel
// on both sides, as two HALF-STRIPS (k 0-31 | 32-63 of the 64-channel chunk; <= 288 rows x 64 B, staged as 24 pieces = 3 per wave, rows
// beyond the strip / outside the image fetch nothing).  The fragment of output pixel o for tap kx is strip row (o / w)(w + 2) + o % w + kx.
//   * LDS: [strip parity][k half] 4 x 24 KB + [B buffer][k half] 4 x 16 KB = 160 KB — all of it;
//   * issue stream per K-tile: ph0 B-hi(t + 1), ph2 B-lo(t + 2) as before; the A stream shrinks to ONE piece of each half of strip s + 1 in
//     phases g = 1, 3, 5 of the strip's twelve (g = 4 kx + ph): 18 instead of 24 DMA instructions per wave and three K-tiles, two per phase like
//     the B events beside them (issued as two bursts of three in g = 1 / 3 the kernel measured 1-2 % slower: the long L segment makes the other
//     wave group wait); a half-strip slot is re-staged >= 2 phases after its last read and first read >= 7 phases after its last piece's issue;
//   * the counted wait keeps the ping-pong kernel's rule — after a phase's issue, at most the pieces of the LAST FOUR issue events may be
//     outstanding — with events of 2 or 0 pieces: vmcnt(4, 6, 6, 8 | 8, 8, 8, 6 | 6, 4, 4, 4) over the 12 phases of a strip;
//   * 6 K-tiles per loop body (tap shift, strip parity and B buffer are compile-time in every phase); the last tiles run the same phases
//     with the "is there anything left to stage" decisions, draining the counter where an issue is skipped.
// Requires: E4T_CONV_S1, chan_major, Cin % 64 == 0, W % 256 == 0 or (256 % W == 0, W >= 16, H % (256 / W) == 0), a K range per split that
// starts and ends on a kernel row (ktiles_per_split % 3 == 0).  Results differ from gemm_pp_kernel's in nothing (same K order, same MFMAs).
// ------------------------------------------------------------------------------------------------
#ifndef PPS_SPREAD
#define PPS_SPREAD 1      // 1: the next strip goes out one piece of each half per phase g = 1, 3, 5 (0: both halves as bursts of three in g = 1 / 3 — 1-2 % slower)
#endif
template <bool GENERAL>
__global__ __launch_bounds__(512) void gemm_pps_kernel(GemmArgs p) {
  constexpr int BM = 256, BN = 256, HK = 32;
  constexpr int QUART = 256 * HK;                        // elements per B quarter
  constexpr int HSTRIP = 384 * HK;                       // elements per half-strip (24 pieces)
  constexpr int A_EL = 4 * HSTRIP;
  constexpr int SMEM = A_EL + 4 * QUART;                 // 160 KB
  static_assert(8 * 64 * (64 + 8) <= SMEM, "write_tile staging must fit");
  __shared__ __attribute__((aligned(16))) bf16_t smem[SMEM];

  const int tid = threadIdx.x;
  const int lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wr = wave >> 2, wc = wave & 3;
  int tile_x, tile_y;
  xcd_tile(tile_x, tile_y, p.group_m);
  const int m0 = tile_y * BM, n0 = tile_x * BN;

  const int nkt = p.K / BK;
  const int bz = blockIdx.z / p.splitk, sz = blockIdx.z - bz * p.splitk;
  if (p.bias) p.bias += bz * p.strideBias;
  if (!p.reduce_batch) {
    if (p.flags & E4T_OUT_F32) p.C = (float*)p.C + bz * p.strideC;
    else p.C = (bf16_t*)p.C + bz * p.strideC;
  }
  const int kt_begin = sz * p.ktiles_per_split;
  int kt_end = kt_begin + p.ktiles_per_split;
  if (kt_end > nkt) kt_end = nkt;

  const int hw = p.Hin * p.Win;
  const int ib = m0 / hw, rem = m0 - ib * hw, y = rem / p.Win, x0 = rem - y * p.Win;
  const int segw = p.Win < 256 ? p.Win : 256, SW = segw + 2, nrows = 256 / segw;
  const __amdgpu_buffer_rsrc_t rs_a = cm_rsrc(p);
  const __amdgpu_buffer_rsrc_t rs_b = uniform_rsrc(p.B, p.b_bytes);
  constexpr unsigned OOB = 0xFFFF0000u;
  const int drow = lane >> 2, dslot = lane & 3;
  // half-strip pieces of this wave: piece q = wave + 8 i = strip rows 16 q .. 16 q + 15
  unsigned a_vo[3];
  int a_inv[3];                                  // bit ky set = this strip row is padding (or no row at all) for kernel row ky
#pragma unroll
  for (int i = 0; i < 3; ++i) {
    const int j = 16 * (wave + 8 * i) + drow;
    const int ry = j / SW, xs = j - ry * SW, ix = x0 - 1 + xs;
    const bool ok = ry < nrows && ix >= 0 && ix < p.Win && m0 < p.M;
    a_vo[i] = (unsigned)(((ry * p.Win + xs) * p.Cin + (dslot ^ ((j >> 2) & 3)) * 8) * 2);
    int m = 0;
#pragma unroll
    for (int ky = 0; ky < 3; ++ky) {
      const int iy = y + ry + ky - 1;
      if (!(ok && iy >= 0 && iy < p.Hin)) m |= 1 << ky;
    }
    a_inv[i] = m;
  }
  unsigned b_vo[2];
#pragma unroll
  for (int j = 0; j < 2; ++j) {
    const int r = wave * 32 + j * 16 + drow;
    const int gn = n0 + r;
    b_vo[j] = gn < p.N ? (unsigned)(((size_t)gn * p.ldb + (dslot ^ ((r >> 2) & 3)) * 8) * 2) : OOB;
  }
  CmWalk wb;                       // B stream: (tap, chunk) of the next K-tile to issue
  wb.init(kt_begin, BK);
  int b_so = 0;
  int a_ky = (kt_begin % 9) / 3, a_chb = (kt_begin / 9) * BK * 2;       // A stream: kernel row and chunk byte offset of the next strip to issue
  auto issue_b = [&](bool hi, bf16_t* dst) __attribute__((always_inline)) {
    if (!hi) b_so = cm_b_so(p, wb);
    else b_so = __builtin_amdgcn_readfirstlane(b_so + HK * 2);
#pragma unroll
    for (int j = 0; j < 2; ++j) buf_dma16(rs_b, b_vo[j], b_so, dst + (wave * 32 + j * 16) * HK);
    if (hi) wb.next(BK);
  };
  auto issue_strip = [&](bool hi, bf16_t* dst) __attribute__((always_inline)) {
    const int iy = y + a_ky - 1;
    // pixel (iy, x0 - 1) of image ib against the lowered base (+ (W + 1) Cin elements): ((ib H + iy + 1) W + x0) Cin  >= 0
    const int so = __builtin_amdgcn_readfirstlane((int)((((long long)ib * p.Hin + iy + 1) * p.Win + x0) * p.Cin * 2) + a_chb + (hi ? HK * 2 : 0));
#pragma unroll
    for (int i = 0; i < 3; ++i)
      buf_dma16(rs_a, a_vo[i] | (unsigned)__builtin_amdgcn_sbfe(a_inv[i], (unsigned)a_ky, 1u), so, dst + (wave + 8 * i) * 512);
    if (hi && ++a_ky == 3) { a_ky = 0; a_chb += BK * 2; }
  };
  // PPS_SPREAD: piece i of BOTH halves of the next strip per call (three calls per strip, in phase 1 of the kx = 0, 1, 2 tiles... see phase())
  auto issue_strip_piece = [&](auto Ic, bool last, bf16_t* dst_lo) __attribute__((always_inline)) {
    constexpr int i = decltype(Ic)::value;
    const int iy = y + a_ky - 1;
    const int so = __builtin_amdgcn_readfirstlane((int)((((long long)ib * p.Hin + iy + 1) * p.Win + x0) * p.Cin * 2) + a_chb);
    const unsigned vo = a_vo[i] | (unsigned)__builtin_amdgcn_sbfe(a_inv[i], (unsigned)a_ky, 1u);
    buf_dma16(rs_a, vo, so, dst_lo + (wave + 8 * i) * 512);
    buf_dma16(rs_a, vo, so + HK * 2, dst_lo + HSTRIP + (wave + 8 * i) * 512);
    if (last && ++a_ky == 3) { a_ky = 0; a_chb += BK * 2; }
  };

  f32x16 acc[4][2];
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

  const int frow = lane & 31, fhi = lane >> 5;
  int a_off[3][4][2], b_off[2][2];     // fragment offsets inside a half-strip (per tap shift) / a quarter (elements)
#pragma unroll
  for (int ks = 0; ks < 2; ++ks) {
#pragma unroll
    for (int kx = 0; kx < 3; ++kx)
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const int o = wr * 128 + i * 32 + frow;
        const int r = (o / segw) * SW + o % segw + kx;
        a_off[kx][i][ks] = r * HK + (((ks * 2 + fhi) ^ ((r >> 2) & 3)) * 8);
      }
#pragma unroll
    for (int j = 0; j < 2; ++j) {
      const int r = wc * 64 + j * 32 + frow;
      b_off[j][ks] = r * HK + (((ks * 2 + fhi) ^ ((r >> 2) & 3)) * 8);
    }
  }
  bf16_t* const Bq = smem + A_EL;      // B quarter (buffer bb, half kh) at Bq + (2 bb + kh) QUART; half-strip (parity sp, half kh) at smem + (2 sp + kh) HSTRIP
  issue_strip(false, smem);
  issue_strip(true, smem + HSTRIP);
  issue_b(false, Bq);
  issue_b(true, Bq + QUART);
  if (kt_begin + 1 < kt_end) issue_b(false, Bq + 2 * QUART);
  wait_vmcnt<0>();                 // (once per workgroup: the counted waits of the loop start from an empty queue)
  __builtin_amdgcn_s_barrier();

  bf16x8 af[2][2], bfr[2][2];
  using I0 = std::integral_constant<int, 0>; using I1 = std::integral_constant<int, 1>;
  using I2 = std::integral_constant<int, 2>; using I3 = std::integral_constant<int, 3>;
  auto phase = [&](auto Bc, auto Pc, auto ALLc, auto KXc, auto SPc, int kt) {
    constexpr int bb = decltype(Bc)::value, ph = decltype(Pc)::value, KX = decltype(KXc)::value, SP = decltype(SPc)::value;
    constexpr bool ALL = decltype(ALLc)::value;      // steady state: everything this phase stages exists
    constexpr int kh = ph >> 1, mh = ph & 1;
    const bf16_t* const qa = smem + (2 * SP + kh) * HSTRIP;
    const bf16_t* const qb = Bq + (2 * bb + kh) * QUART;
    // ---- L segment ----
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
      for (int ks = 0; ks < 2; ++ks) af[i][ks] = *(const bf16x8*)(qa + a_off[KX][mh * 2 + i][ks]);
    if (mh == 0) {
#pragma unroll
      for (int j = 0; j < 2; ++j)
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) bfr[j][ks] = *(const bf16x8*)(qb + b_off[j][ks]);
    }
    bool staged = true;              // phases that issue nothing by design keep the pattern's count
    if (ph == 0)      { staged = ALL || kt + 1 < kt_end; if (staged) issue_b(true, Bq + (2 * (bb ^ 1) + 1) * QUART); }
    else if (ph == 2) { staged = ALL || kt + 2 < kt_end; if (staged) issue_b(false, Bq + (2 * bb + 0) * QUART); }
#if PPS_SPREAD
    // one piece of each half of the next strip in phases g = 1, 3, 5 of the strip's twelve (both halves land >= 5 events before their first read at g = 12 / 14)
    else if (4 * KX + ph == 1) { staged = ALL || kt + 3 < kt_end; if (staged) issue_strip_piece(I0{}, false, smem + 2 * (SP ^ 1) * HSTRIP); }
    else if (4 * KX + ph == 3) { staged = ALL || kt + 3 < kt_end; if (staged) issue_strip_piece(I1{}, false, smem + 2 * (SP ^ 1) * HSTRIP); }
    else if (4 * KX + ph == 5) { staged = ALL || kt + 2 < kt_end; if (staged) issue_strip_piece(I2{}, true, smem + 2 * (SP ^ 1) * HSTRIP); }
    constexpr int G_ = 4 * KX + ph;
    constexpr int NW_ = G_ == 0 ? 4 : G_ <= 2 ? 6 : G_ <= 6 ? 8 : G_ <= 8 ? 6 : 4;
#else
    else if (KX == 0) { staged = ALL || kt + 3 < kt_end; if (staged) issue_strip(ph == 3, smem + (2 * (SP ^ 1) + kh) * HSTRIP); }
    // at most the pieces of the last four issue events outstanding (2 per B quarter, 3 per half-strip, 0 where nothing is issued)
    constexpr int NW_ = KX == 0 ? (ph == 0 ? 4 : ph == 3 ? 10 : 7) : KX == 1 ? (ph == 0 ? 10 : ph == 3 ? 4 : 7) : 4;
#endif
    if (ALL || staged) wait_vmcnt<NW_>(); else wait_vmcnt<0>();
    __builtin_amdgcn_sched_barrier(0);
    __builtin_amdgcn_s_barrier();
    __builtin_amdgcn_sched_barrier(0);
    // ---- M segment ----
    __builtin_amdgcn_s_setprio(1);
#pragma unroll
    for (int ks = 0; ks < 2; ++ks)
#pragma unroll
      for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
          acc[mh * 2 + i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[i][ks], bfr[j][ks], acc[mh * 2 + i][j], 0, 0, 0);
    __builtin_amdgcn_s_setprio(0);
    __builtin_amdgcn_sched_barrier(0);
    __builtin_amdgcn_s_barrier();
    __builtin_amdgcn_sched_barrier(0);
  };
  // K-tile i of a 6-tile body: B buffer i & 1, tap shift i % 3, strip parity (i / 3) & 1
  auto ktile = [&](auto Bc, auto KXc, auto SPc, auto ALLc, int kt) __attribute__((always_inline)) {
    phase(Bc, I0{}, ALLc, KXc, SPc, kt); phase(Bc, I1{}, ALLc, KXc, SPc, kt); phase(Bc, I2{}, ALLc, KXc, SPc, kt); phase(Bc, I3{}, ALLc, KXc, SPc, kt);
  };

  if (wr == 1) __builtin_amdgcn_s_barrier();        // group 1 runs one barrier interval behind group 0
  {
    using YES = std::true_type; using NO = std::false_type;
    int kt = kt_begin;
    for (; kt + 8 <= kt_end; kt += 6) {       // steady state: everything the six tiles stage (up to B-lo of tile kt + 7, the strip of kt + 6) exists
      ktile(I0{}, I0{}, I0{}, YES{}, kt);     ktile(I1{}, I1{}, I0{}, YES{}, kt + 1); ktile(I0{}, I2{}, I0{}, YES{}, kt + 2);
      ktile(I1{}, I0{}, I1{}, YES{}, kt + 3); ktile(I0{}, I1{}, I1{}, YES{}, kt + 4); ktile(I1{}, I2{}, I1{}, YES{}, kt + 5);
    }
    if (kt < kt_end) { ktile(I0{}, I0{}, I0{}, NO{}, kt); ktile(I1{}, I1{}, I0{}, NO{}, kt + 1); ktile(I0{}, I2{}, I0{}, NO{}, kt + 2); kt += 3; }
    if (kt < kt_end) { ktile(I1{}, I0{}, I1{}, NO{}, kt); ktile(I0{}, I1{}, I1{}, NO{}, kt + 1); ktile(I1{}, I2{}, I1{}, NO{}, kt + 2); kt += 3; }
  }
  if (wr == 0) __builtin_amdgcn_s_barrier();
  __syncthreads();   // every fragment read and every DMA is done before the epilogue reuses the LDS
  write_tile<64, 64, 2, 2, GENERAL>(p, *(f32x16(*)[2][2])(acc + 0), wave_stage<64, 64>(smem, wave), lane, m0 + wr * 128, n0 + wc * 64);
  __syncthreads();
  write_tile<64, 64, 2, 2, GENERAL>(p, *(f32x16(*)[2][2])(acc + 2), wave_stage<64, 64>(smem, wave), lane, m0 + wr * 128 + 64, n0 + wc * 64);
}
// may a 256 x 256 ping-pong conv launch go to gemm_pps_kernel?
static bool conv_pps_ok(const GemmArgs& p, int batch) {
  static const bool on = CONV_STRIP && getenv("E4T_CONV_NOSTRIP") == nullptr;          // A/B switch
  const bool rows = p.Win % 256 == 0 || (p.Win >= 16 && 256 % p.Win == 0 && p.Hin % (256 / p.Win) == 0);
  return on && p.mode == E4T_CONV_S1 && p.chan_major && batch == 1 && rows && p.Wout == p.Win && p.Hout == p.Hin && p.Cin % 64 == 0 &&
         p.K == 9 * p.Cin && p.ktiles_per_split % 3 == 0 && p.M % 256 == 0;
}

// ------------------------------------------------------------------------------------------------
// 256 x 320 ping-pong, phases = (k half) x (16-wide k step)      [round 3; the template also builds BN = 256, measured = gemm_pp_kernel]
//
// The same machine as gemm_pp_kernel — two groups of four waves alternating an LDS/DMA segment with an MFMA segment, operand
// quarters of 32 k moved by LDS-DMA, counted vmcnt — with the K-tile cut along K instead of along the wave tile's rows: a phase is
// one 16-wide k step over the WHOLE wave tile.  Every phase then reads the same number of fragments (FM + FN ds_read_b128 for
// FM x FN MFMAs) instead of alternating 8 / 4, holds one k step of fragments in registers instead of two (24 instead of 32 VGPRs
// at BN = 256), and — what it was written for — makes a 256 x 320 tile fit: 8 waves = 4 (rows) x 2 (columns), wave tile 64 x 160
// (2 x 5 accumulators = 160 VGPRs + 28 fragment registers), 10 MFMAs per phase.  A 320-wide tile covers the UNet's 320 / 640 /
// 960 / 1280 / 1920 / 2560-channel layers without N padding at 0.93 KB of LDS traffic per MFMA (fragment reads + DMA writes), where
// the 128 x 160 tile moves 1.69 KB (the GEMM family is bound by LDS traffic, not by the DMA path or the MFMA rate).
//   * quarters: A = 256 rows x 32 k (2 DMA instructions per wave), B = BN rows x 32 k (BN = 320: 20 pieces, 3 per wave in waves
//     0-3, 2 in waves 4-7); two buffers x [A-lo | B-lo | A-hi | B-hi] = 128 / 144 KiB;
//   * phases 0, 1 read the lo quarters, 2, 3 the hi quarters.  A slot is re-staged no earlier than two phases after its last read
//     (the trailing group's reads), so the issue stream runs  ph1: B-hi(t+1)  ph2: A-hi(t+1)  ph3: B-lo(t+2)  ph0': A-lo(t+2)
//     (A-lo of tile t+2 is issued in the first phase of tile t+1); every quarter is first read >= 3 phases after its issue, and
//     "at most the 3 newest quarters outstanding" after each issue is exactly what the next phase needs landed — 2 A + 1 B quarters
//     after an even phase, 1 A + 2 B after an odd one (instruction counts differ per wave at BN = 320).
// Requires K % 64 == 0.  Same epilogue code as every other kernel: results are bit-identical to theirs.
// ------------------------------------------------------------------------------------------------
template <int MODE, int BN, bool GENERAL>
__global__ __launch_bounds__(512) void gemm_pq_kernel(GemmArgs p) {
  constexpr int BM = 256, HK = 32;
  constexpr int WR = BN == 320 ? 4 : 2, WC = 8 / WR;
  constexpr int WM = BM / WR, WN = BN / WC;                  // 128 x 64 | 64 x 160
  constexpr int FM = WM / 32, FN = WN / 32;                  // 4 x 2 | 2 x 5
  constexpr int QA = BM * HK, QB = BN * HK;                  // elements per A / B quarter
  constexpr int HALF = QA + QB, BUF = 2 * HALF;              // [A | B] of one k half; one buffer = lo half + hi half
  constexpr int NBP = BN / 16;                               // B pieces per quarter: 16 | 20
  constexpr int NBJ = (NBP + 7) / 8;                         // per wave: 2 | 3 (the last one only in waves < NBP - 8 * (NBJ - 1))
  constexpr int NBLAST = NBP - 8 * (NBJ - 1);                // waves that carry NBJ pieces: 8 | 4
  constexpr int ESTG = 8 * 32 * (WN + 8);                    // epilogue staging: 8 waves x 32 rows
  static_assert(BN == 256 || BN == 320, "tile width");
  static_assert(2 * BUF * 2 <= 160 * 1024 && ESTG <= 2 * BUF, "LDS budget");
  __shared__ __attribute__((aligned(16))) bf16_t smem[2 * BUF];

  const int tid = threadIdx.x;
  const int lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int grp = wave >> 2;
  const int wr = wave / WC, wc = wave % WC;
  int tile_x, tile_y;
  xcd_tile(tile_x, tile_y, p.group_m);
  int m0 = tile_y * BM;
  const int n0 = tile_x * BN;
  const int mrows = p.M - m0;                       // valid (logical) rows of this tile
  if (p.panel_rows) m0 = (m0 / p.panel_rows) * p.panel_stride + p.panel_off + m0 % p.panel_rows;      // physical first row (panel_rows % 256 == 0)

  const int nkt = p.K / BK;
  const int bz = blockIdx.z / p.splitk, sz = blockIdx.z - bz * p.splitk;
  p.A += bz * p.strideA;
  if (p.A2) p.A2 += bz * p.strideA;
  p.B += bz * p.strideB;
  if (p.bias) p.bias += bz * p.strideBias;
  if (!p.reduce_batch) {
    if (p.flags & E4T_OUT_F32) p.C = (float*)p.C + bz * p.strideC;
    else p.C = (bf16_t*)p.C + bz * p.strideC;
  }
  const int kt_begin = sz * p.ktiles_per_split;
  int kt_end = kt_begin + p.ktiles_per_split;
  if (kt_end > nkt) kt_end = nkt;

  const bool cm = MODE != 0 && p.chan_major;          // channel-chunk-major K order (gemm_common.h, cm_step)
  const __amdgpu_buffer_rsrc_t rs_a = cm ? cm_rsrc(p) : __builtin_amdgcn_make_buffer_rsrc((void*)p.A, 0, (int)p.a_bytes, 0x00020000);
  const __amdgpu_buffer_rsrc_t rs_a2 = __builtin_amdgcn_make_buffer_rsrc((void*)(p.A2 ? p.A2 : p.A), 0, (int)(p.A2 ? p.a2_bytes : p.a_bytes), 0x00020000);
  const __amdgpu_buffer_rsrc_t rs_b = __builtin_amdgcn_make_buffer_rsrc((void*)p.B, 0, (int)p.b_bytes, 0x00020000);
  constexpr unsigned OOB = 0xFFFF0000u;
  // DMA: one wave-instruction = 16 rows x 64 B; wave w feeds A rows 32w + 16j + (lane >> 2), j = 0, 1, and B pieces w + 8j
  const int drow = lane >> 2, dslot = lane & 3;
  long long a_base[2];
  int a_oy[2], a_ox[2], a_kc[2];
  bool a_ok[2];
  unsigned b_vo[NBJ];
#pragma unroll
  for (int j = 0; j < 2; ++j) {
    const int r = wave * 32 + j * 16 + drow;
    a_kc[j] = (dslot ^ ((r >> 2) & 3)) * 8;
    const int gr = m0 + r;
    a_ok[j] = r < mrows;
    if (MODE == 0) {
      a_base[j] = (long long)gr; a_oy[j] = a_ox[j] = 0;
    } else {
      const int hw = p.Hout * p.Wout;
      const int b = gr / hw;
      const int rem = gr - b * hw;
      a_oy[j] = rem / p.Wout;
      a_ox[j] = rem - a_oy[j] * p.Wout;
      a_base[j] = (long long)b * p.Hin * p.Win;
    }
  }
#pragma unroll
  for (int j = 0; j < NBJ; ++j) {
    const int r = (wave + 8 * j) * 16 + drow;
    const int kc = (dslot ^ ((r >> 2) & 3)) * 8;
    const int gn = n0 + r;
    b_vo[j] = (r < BN && gn < p.N) ? (unsigned)(((size_t)gn * p.ldb + kc) * 2) : OOB;
  }
  unsigned a_vo[2];
  if (cm) {                        // a_vo = the output position's own pixel (all taps), a_oy = inverted 9-bit tap validity mask
#pragma unroll
    for (int j = 0; j < 2; ++j) {
      a_vo[j] = cm_center(p, a_base[j], a_oy[j], a_ox[j], a_kc[j]);
      a_oy[j] = cm_inv_mask(p, a_ok[j], a_oy[j], a_ox[j]);
    }
  }
  unsigned a_eff[2] = {0u, 0u};    // cm: the offsets of the K-tile whose quarters are being issued
  CmWalk wa, wb;                   // cm: one walker per operand stream (A and B are issued at different times)
  wa.init(kt_begin, BK); wb = wa;
  const int cm_table = cm ? cm_tap_table(p, lane) : 0;
  int a_so = 0, b_so = 0;
  bool a_second = false;
  auto place_a = [&](int k0) {
    if (MODE == 0) {
      int ld = p.lda, koff = k0;
      a_second = k0 >= p.K1;
      if (a_second) { ld = p.lda2; koff = k0 - p.K1; }
      a_so = __builtin_amdgcn_readfirstlane(koff * 2);          // (wave-uniform by construction; keeps the offset in an SGPR for the compiler)
#pragma unroll
      for (int j = 0; j < 2; ++j) a_vo[j] = a_ok[j] ? (unsigned)((a_base[j] * ld + a_kc[j]) * 2) : OOB;
    } else {
      const int tap = k0 / p.Cin;
      const int ci0 = k0 - tap * p.Cin;
      const int ky = tap / 3, kx = tap - ky * 3;
      a_so = __builtin_amdgcn_readfirstlane(ci0 * 2);
#pragma unroll
      for (int j = 0; j < 2; ++j) {
        int iy, ix;
        bool ok = a_ok[j];
        if (p.mode == E4T_CONV_S1) {
          iy = a_oy[j] + ky - 1; ix = a_ox[j] + kx - 1;
          ok = ok && iy >= 0 && iy < p.Hin && ix >= 0 && ix < p.Win;
        } else if (p.mode == E4T_CONV_S2) {
          iy = 2 * a_oy[j] + ky - 1; ix = 2 * a_ox[j] + kx - 1;
          ok = ok && iy >= 0 && iy < p.Hin && ix >= 0 && ix < p.Win;
        } else if (p.mode == E4T_CONV_UP2) {
          iy = a_oy[j] + ky - 1; ix = a_ox[j] + kx - 1;
          ok = ok && iy >= 0 && iy < 2 * p.Hin && ix >= 0 && ix < 2 * p.Win;
          iy >>= 1; ix >>= 1;
        } else if (p.mode == E4T_CONV_S2A) {
          iy = 2 * a_oy[j] + ky; ix = 2 * a_ox[j] + kx;
          ok = ok && iy < p.Hin && ix < p.Win;
        } else {
          const int sy = a_oy[j] + ky - 1, sx = a_ox[j] + kx - 1;
          ok = ok && sy >= 0 && sx >= 0 && !(sy & 1) && !(sx & 1);
          iy = sy >> 1; ix = sx >> 1;
          ok = ok && iy < p.Hin && ix < p.Win;
        }
        a_vo[j] = ok ? (unsigned)(((a_base[j] + (long long)iy * p.Win + ix) * p.Cin + a_kc[j]) * 2) : OOB;
      }
    }
  };
  // the A and B streams are each issued in increasing k: lo(t), hi(t), lo(t+1), ...
  auto issue_a = [&](int kt, bool hi, bf16_t* dst) __attribute__((always_inline)) {
    if (cm) {                        // lo(t), hi(t), lo(t+1), ...: the walker advances after each hi
      if (!hi) {
        a_so = cm_a_so(cm_table, wa);
#pragma unroll
        for (int j = 0; j < 2; ++j) a_eff[j] = cm_row_off(a_vo[j], a_oy[j], wa);
      } else {
        a_so = __builtin_amdgcn_readfirstlane(a_so + HK * 2);
      }
#pragma unroll
      for (int j = 0; j < 2; ++j) buf_dma16(rs_a, a_eff[j], a_so, dst + (wave * 32 + j * 16) * HK);
      if (hi) wa.next(BK);
      return;
    }
    const int k0 = kt * BK;
    const bool fresh = !hi && (kt == kt_begin || (MODE == 0 ? k0 == p.K1 : (k0 % p.Cin) == 0));
    if (fresh) place_a(k0);
    else a_so = __builtin_amdgcn_readfirstlane(a_so + HK * 2);
#pragma unroll
    for (int j = 0; j < 2; ++j)
      buf_dma16(a_second ? rs_a2 : rs_a, a_vo[j], a_so, dst + (wave * 32 + j * 16) * HK);
  };
  auto issue_b = [&](int kt, bool hi, bf16_t* dst) __attribute__((always_inline)) {
    if (cm && !hi) b_so = cm_b_so(p, wb);
    else if (!cm && !hi && kt == kt_begin) b_so = kt * BK * 2;
    else b_so = __builtin_amdgcn_readfirstlane(b_so + HK * 2);
#pragma unroll
    for (int j = 0; j < NBJ; ++j)
      if (j < NBJ - 1 || wave < NBLAST) buf_dma16(rs_b, b_vo[j], b_so, dst + ((wave + 8 * j) * 16) * HK);
    if (cm && hi) wb.next(BK);
  };
  // at most the 3 newest quarters of this wave outstanding: 2 A + 1 B after an even phase, 1 A + 2 B after an odd one
  constexpr int NBW_HI = NBJ, NBW_LO = NBJ - 1;
  auto wait3 = [&](auto ODDc) {
    constexpr bool ODD = decltype(ODDc)::value;
    if (NBLAST == 8 || wave < NBLAST) wait_vmcnt<(ODD ? 2 + 2 * NBW_HI : 4 + NBW_HI)>();
    else wait_vmcnt<(ODD ? 2 + 2 * NBW_LO : 4 + NBW_LO)>();
  };

  f32x16 acc[FM][FN];
#pragma unroll
  for (int i = 0; i < FM; ++i)
#pragma unroll
    for (int j = 0; j < FN; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

  const int frow = lane & 31, fhi = lane >> 5;
  int a_off[FM][2], b_off[FN][2];     // fragment offsets inside a quarter (elements), [block][k step of the half]
#pragma unroll
  for (int ks = 0; ks < 2; ++ks) {
#pragma unroll
    for (int i = 0; i < FM; ++i) {
      const int r = wr * WM + i * 32 + frow;
      a_off[i][ks] = r * HK + (((ks * 2 + fhi) ^ ((r >> 2) & 3)) * 8);
    }
#pragma unroll
    for (int j = 0; j < FN; ++j) {
      const int r = wc * WN + j * 32 + frow;
      b_off[j][ks] = QA + r * HK + (((ks * 2 + fhi) ^ ((r >> 2) & 3)) * 8);
    }
  }
  // buffer b = smem + b * BUF: [lo: A | B][hi: A | B].  Prologue: lo(t0), hi(t0), B-lo(t0+1), A-lo(t0+1) — the stream position the
  // first phase expects (its own issue is A-lo(t+2)... of the NEXT tile: see the phase schedule).
  issue_b(kt_begin, false, smem + QA);
  issue_a(kt_begin, false, smem);
  issue_b(kt_begin, true, smem + HALF + QA);
  issue_a(kt_begin, true, smem + HALF);
  if (kt_begin + 1 < kt_end) {
    issue_b(kt_begin + 1, false, smem + BUF + QA);
    issue_a(kt_begin + 1, false, smem + BUF);
  }
  wait_vmcnt<0>();                 // (one-off: the first K-tile and the next one's lo half)
  __builtin_amdgcn_s_barrier();

  auto phase = [&](auto Bc, auto Pc, int kt) {
    constexpr int b = decltype(Bc)::value, ph = decltype(Pc)::value;
    constexpr int kh = ph >> 1, ks = ph & 1;
    bf16_t* const buf = smem + b * BUF;
    bf16_t* const other = smem + (b ^ 1) * BUF;
    const bf16_t* const q = buf + kh * HALF;
    bf16x8 af[FM], bfr[FN];
    // ---- L segment ----
#pragma unroll
    for (int i = 0; i < FM; ++i) af[i] = *(const bf16x8*)(q + a_off[i][ks]);
#pragma unroll
    for (int j = 0; j < FN; ++j) bfr[j] = *(const bf16x8*)(q + b_off[j][ks]);
    bool staged;
    if (ph == 0)      { staged = kt + 1 < kt_end && kt > kt_begin; if (staged) issue_a(kt + 1, false, other); }       // A-lo(t+1) (the prologue issued the first one)
    else if (ph == 1) { staged = kt + 1 < kt_end; if (staged) issue_b(kt + 1, true, other + HALF + QA); }              // B-hi(t+1)
    else if (ph == 2) { staged = kt + 1 < kt_end; if (staged) issue_a(kt + 1, true, other + HALF); }                   // A-hi(t+1)
    else              { staged = kt + 2 < kt_end; if (staged) issue_b(kt + 2, false, buf + QA); }                      // B-lo(t+2)
    if (staged) wait3(std::integral_constant<bool, (ph & 1) != 0>{}); else wait_vmcnt<0>();
    __builtin_amdgcn_sched_barrier(0);
    __builtin_amdgcn_s_barrier();
    __builtin_amdgcn_sched_barrier(0);
    // ---- M segment ----
    __builtin_amdgcn_s_setprio(1);
#pragma unroll
    for (int i = 0; i < FM; ++i)
#pragma unroll
      for (int j = 0; j < FN; ++j)
        acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[i], bfr[j], acc[i][j], 0, 0, 0);
    __builtin_amdgcn_s_setprio(0);
    __builtin_amdgcn_sched_barrier(0);
    __builtin_amdgcn_s_barrier();
    __builtin_amdgcn_sched_barrier(0);
  };
  using I0 = std::integral_constant<int, 0>; using I1 = std::integral_constant<int, 1>;
  using I2 = std::integral_constant<int, 2>; using I3 = std::integral_constant<int, 3>;

  if (grp == 1) __builtin_amdgcn_s_barrier();        // group 1 runs one barrier interval behind group 0
  {
    int kt = kt_begin;
    for (; kt + 2 <= kt_end; kt += 2) {
      phase(I0{}, I0{}, kt); phase(I0{}, I1{}, kt); phase(I0{}, I2{}, kt); phase(I0{}, I3{}, kt);
      phase(I1{}, I0{}, kt + 1); phase(I1{}, I1{}, kt + 1); phase(I1{}, I2{}, kt + 1); phase(I1{}, I3{}, kt + 1);
    }
    if (kt < kt_end) { phase(I0{}, I0{}, kt); phase(I0{}, I1{}, kt); phase(I0{}, I2{}, kt); phase(I0{}, I3{}, kt); }
  }
  if (grp == 0) __builtin_amdgcn_s_barrier();
  __syncthreads();   // every fragment read and every DMA is done before the epilogue reuses the LDS
  if (p.panel_rows) p.M = m0 + min(mrows, BM);      // the epilogue bounds PHYSICAL rows (never with split-K: p.M is the slab stride there)
  // 32-row slices of the wave tile: the staging of 8 waves x 32 x (WN + 8) fits the operand buffers, the unrolled epilogue stays small
  // (compile-time row index: a runtime-indexed accumulator array is placed in scratch memory — 704 bytes per lane written and read back
  // through HBM cost ~55 us per tile in the first version of this kernel)
  auto slice = [&](auto Ic) {
    constexpr int i = decltype(Ic)::value;
    write_tile<32, WN, 1, FN, GENERAL>(p, *(f32x16(*)[1][FN])(&acc[i][0]), wave_stage<32, WN>(smem, wave), lane, m0 + wr * WM + i * 32, n0 + wc * WN);
  };
  slice(I0{});
  __syncthreads();
  slice(I1{});
  if constexpr (FM == 4) {
    __syncthreads();
    slice(I2{});
    __syncthreads();
    slice(I3{});
  }
  if constexpr (MODE == 0) {
    if (p.tail_rows) gemm_tail<8, GENERAL>(p, (float*)smem, wave, lane);
  }
}

#ifdef E4T_EXPERIMENTAL
// ------------------------------------------------------------------------------------------------
// 512 x 128 ("tall") ping-pong tile: the same phase machine with the operand roles exchanged, for outputs that are only
// 128 columns wide (the VAE's 128-channel 3x3 convs at 512^2: M = 4.2 M rows, N = 128 — a 256-wide tile would be half empty
// and the 128 x 128 tile reaches 680 TF/s there).
//   * 8 waves = 2 groups (wr, 256 rows each) x 4 row slices (wc, 64 rows), wave tile 64 x 128 (2 x 4 accumulators);
//   * phases = (k 0-31 | 32-63) x (columns 0-63 | 64-127): B fragments are read every phase, A fragments when nh == 0;
//   * quarters: A = 512 rows x 32 k = 32 KiB (4 DMA instructions per wave), B = 128 rows x 32 k = 8 KiB (1 instruction);
//     two buffers x two k-halves x (A + B) = 160 KiB = the whole LDS of a CU.  Any 4 consecutive quarters are 2 A + 2 B =
//     10 instructions per wave, hence vmcnt(10) where the square kernel has vmcnt(8).  Issue order per K-tile:
//     A-hi(t+1), B-hi(t+1), A-lo(t+2), B-lo(t+2) in phases 0..3 — every slot is re-staged 2 phases after its last read.
// ------------------------------------------------------------------------------------------------
template <int MODE>
__global__ __launch_bounds__(512) void gemm_pt_kernel(GemmArgs p) {
  constexpr int BM = 512, BN = 128, HK = 32;
  constexpr int QA = BM * HK, QB = BN * HK;              // elements per A / B quarter
  constexpr int HALF = QA + QB;                          // one k-half of a buffer: [A | B]
  constexpr int SMEM = 4 * HALF;                         // 81920 elements = 160 KiB
  static_assert(8 * 64 * (64 + 8) <= SMEM, "epilogue staging must fit");
  __shared__ __attribute__((aligned(16))) bf16_t smem[SMEM];

  const int tid = threadIdx.x;
  const int lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wr = wave >> 2, wc = wave & 3;
  int tile_x, tile_y;
  xcd_tile(tile_x, tile_y, p.group_m);
  const int m0 = tile_y * BM, n0 = tile_x * BN;

  const int nkt = p.K / BK;
  const int bz = blockIdx.z / p.splitk, sz = blockIdx.z - bz * p.splitk;
  p.A += bz * p.strideA;
  if (p.A2) p.A2 += bz * p.strideA;
  p.B += bz * p.strideB;
  if (p.bias) p.bias += bz * p.strideBias;
  if (!p.reduce_batch) {
    if (p.flags & E4T_OUT_F32) p.C = (float*)p.C + bz * p.strideC;
    else p.C = (bf16_t*)p.C + bz * p.strideC;
  }
  const int kt_begin = sz * p.ktiles_per_split;
  int kt_end = kt_begin + p.ktiles_per_split;
  if (kt_end > nkt) kt_end = nkt;

  const __amdgpu_buffer_rsrc_t rs_a = __builtin_amdgcn_make_buffer_rsrc((void*)p.A, 0, (int)p.a_bytes, 0x00020000);
  const __amdgpu_buffer_rsrc_t rs_a2 = __builtin_amdgcn_make_buffer_rsrc((void*)(p.A2 ? p.A2 : p.A), 0, (int)(p.A2 ? p.a2_bytes : p.a_bytes), 0x00020000);
  const __amdgpu_buffer_rsrc_t rs_b = __builtin_amdgcn_make_buffer_rsrc((void*)p.B, 0, (int)p.b_bytes, 0x00020000);
  constexpr unsigned OOB = 0xFFFF0000u;
  // DMA: one wave-instruction = 16 rows x 64 B.  A: wave w feeds quarter rows 64w + 16j + (lane >> 2), j = 0..3; B: rows 16w + (lane >> 2)
  const int drow = lane >> 2, dslot = lane & 3;
  long long a_base[4];
  int a_oy[4], a_ox[4], a_kc[4];
  bool a_ok[4];
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    const int r = wave * 64 + j * 16 + drow;
    a_kc[j] = (dslot ^ ((r >> 2) & 3)) * 8;
    const int gr = m0 + r;
    a_ok[j] = gr < p.M;
    if (MODE == 0) {
      a_base[j] = (long long)gr; a_oy[j] = a_ox[j] = 0;
    } else {
      const int hw = p.Hout * p.Wout;
      const int b = gr / hw;
      const int rem = gr - b * hw;
      a_oy[j] = rem / p.Wout;
      a_ox[j] = rem - a_oy[j] * p.Wout;
      a_base[j] = (long long)b * p.Hin * p.Win;
    }
  }
  unsigned b_vo;
  {
    const int r = wave * 16 + drow;
    const int kc = (dslot ^ ((r >> 2) & 3)) * 8;
    const int gn = n0 + r;
    b_vo = gn < p.N ? (unsigned)(((size_t)gn * p.ldb + kc) * 2) : OOB;
  }
  unsigned a_vo[4];
  int a_so = 0, b_so = 0;
  bool a_second = false;
  auto place_a = [&](int k0) {
    if (MODE == 0) {
      int ld = p.lda, koff = k0;
      a_second = k0 >= p.K1;
      if (a_second) { ld = p.lda2; koff = k0 - p.K1; }
      a_so = __builtin_amdgcn_readfirstlane(koff * 2);          // (wave-uniform by construction; keeps the offset in an SGPR for the compiler)
#pragma unroll
      for (int j = 0; j < 4; ++j) a_vo[j] = a_ok[j] ? (unsigned)((a_base[j] * ld + a_kc[j]) * 2) : OOB;
    } else {
      const int tap = k0 / p.Cin;
      const int ci0 = k0 - tap * p.Cin;
      const int ky = tap / 3, kx = tap - ky * 3;
      a_so = __builtin_amdgcn_readfirstlane(ci0 * 2);
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        int iy, ix;
        bool ok = a_ok[j];
        if (p.mode == E4T_CONV_S1) {
          iy = a_oy[j] + ky - 1; ix = a_ox[j] + kx - 1;
          ok = ok && iy >= 0 && iy < p.Hin && ix >= 0 && ix < p.Win;
        } else if (p.mode == E4T_CONV_S2) {
          iy = 2 * a_oy[j] + ky - 1; ix = 2 * a_ox[j] + kx - 1;
          ok = ok && iy >= 0 && iy < p.Hin && ix >= 0 && ix < p.Win;
        } else if (p.mode == E4T_CONV_UP2) {
          iy = a_oy[j] + ky - 1; ix = a_ox[j] + kx - 1;
          ok = ok && iy >= 0 && iy < 2 * p.Hin && ix >= 0 && ix < 2 * p.Win;
          iy >>= 1; ix >>= 1;
        } else if (p.mode == E4T_CONV_S2A) {
          iy = 2 * a_oy[j] + ky; ix = 2 * a_ox[j] + kx;
          ok = ok && iy < p.Hin && ix < p.Win;
        } else {
          const int sy = a_oy[j] + ky - 1, sx = a_ox[j] + kx - 1;
          ok = ok && sy >= 0 && sx >= 0 && !(sy & 1) && !(sx & 1);
          iy = sy >> 1; ix = sx >> 1;
          ok = ok && iy < p.Hin && ix < p.Win;
        }
        a_vo[j] = ok ? (unsigned)(((a_base[j] + (long long)iy * p.Win + ix) * p.Cin + a_kc[j]) * 2) : OOB;
      }
    }
  };
  auto issue_a = [&](int kt, bool hi, bf16_t* dst) __attribute__((always_inline)) {
    const int k0 = kt * BK;
    const bool fresh = !hi && (kt == kt_begin || (MODE == 0 ? k0 == p.K1 : (k0 % p.Cin) == 0));
    if (fresh) place_a(k0);
    else a_so = __builtin_amdgcn_readfirstlane(a_so + HK * 2);
#pragma unroll
    for (int j = 0; j < 4; ++j)
      buf_dma16(a_second ? rs_a2 : rs_a, a_vo[j], a_so, dst + (wave * 64 + j * 16) * HK);
  };
  auto issue_b = [&](int kt, bool hi, bf16_t* dst) __attribute__((always_inline)) {
    if (!hi && kt == kt_begin) b_so = kt * BK * 2;
    else b_so = __builtin_amdgcn_readfirstlane(b_so + HK * 2);
    buf_dma16(rs_b, b_vo, b_so, dst + (wave * 16) * HK);
  };

  f32x16 acc[2][2][2];          // [column half][row block][column block]
#pragma unroll
  for (int h = 0; h < 2; ++h)
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
      for (int j = 0; j < 2; ++j)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[h][i][j][r] = 0.f;

  const int frow = lane & 31, fhi = lane >> 5;
  int a_off[2][2], b_off[4][2];     // fragment offsets inside a quarter (elements), [block][k-step of the half]
#pragma unroll
  for (int ks = 0; ks < 2; ++ks) {
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      const int r = wr * 256 + wc * 64 + i * 32 + frow;
      a_off[i][ks] = r * HK + (((ks * 2 + fhi) ^ ((r >> 2) & 3)) * 8);
    }
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const int r = j * 32 + frow;
      b_off[j][ks] = r * HK + (((ks * 2 + fhi) ^ ((r >> 2) & 3)) * 8);
    }
  }
  // buffer b = smem + 2b * HALF: [lo: A | B][hi: A | B]
  issue_a(kt_begin, false, smem);
  issue_b(kt_begin, false, smem + QA);
  issue_a(kt_begin, true, smem + HALF);
  issue_b(kt_begin, true, smem + HALF + QA);
  if (kt_begin + 1 < kt_end) {
    issue_a(kt_begin + 1, false, smem + 2 * HALF);
    issue_b(kt_begin + 1, false, smem + 2 * HALF + QA);
    wait_vmcnt<10>();              // 15 issued: the two lo quarters of the first K-tile have landed
  } else {
    wait_vmcnt<0>();
  }
  __builtin_amdgcn_s_barrier();

  bf16x8 af[2][2], bfr[2][2];
  auto phase = [&](auto Bc, auto Pc, int kt) {
    constexpr int b = decltype(Bc)::value, ph = decltype(Pc)::value;
    constexpr int kh = ph >> 1, nh = ph & 1;
    bf16_t* const buf = smem + b * 2 * HALF;
    bf16_t* const other = smem + (b ^ 1) * 2 * HALF;
    const bf16_t* const qa = buf + kh * HALF;
    const bf16_t* const qb = qa + QA;
    // ---- L segment ----
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int ks = 0; ks < 2; ++ks) bfr[j][ks] = *(const bf16x8*)(qb + b_off[nh * 2 + j][ks]);
    if (nh == 0) {
#pragma unroll
      for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) af[i][ks] = *(const bf16x8*)(qa + a_off[i][ks]);
    }
    bool staged;
    if (ph == 0)      { staged = kt + 1 < kt_end; if (staged) issue_a(kt + 1, true, other + HALF); }
    else if (ph == 1) { staged = kt + 1 < kt_end; if (staged) issue_b(kt + 1, true, other + HALF + QA); }
    else if (ph == 2) { staged = kt + 2 < kt_end; if (staged) issue_a(kt + 2, false, buf); }
    else              { staged = kt + 2 < kt_end; if (staged) issue_b(kt + 2, false, buf + QA); }
    if (staged) wait_vmcnt<10>(); else wait_vmcnt<0>();
    __builtin_amdgcn_sched_barrier(0);
    __builtin_amdgcn_s_barrier();
    __builtin_amdgcn_sched_barrier(0);
    // ---- M segment ----
    __builtin_amdgcn_s_setprio(1);
#pragma unroll
    for (int ks = 0; ks < 2; ++ks)
#pragma unroll
      for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
          acc[nh][i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[i][ks], bfr[j][ks], acc[nh][i][j], 0, 0, 0);
    __builtin_amdgcn_s_setprio(0);
    __builtin_amdgcn_sched_barrier(0);
    __builtin_amdgcn_s_barrier();
    __builtin_amdgcn_sched_barrier(0);
  };
  using I0 = std::integral_constant<int, 0>; using I1 = std::integral_constant<int, 1>;
  using I2 = std::integral_constant<int, 2>; using I3 = std::integral_constant<int, 3>;

  if (wr == 1) __builtin_amdgcn_s_barrier();        // group 1 runs one barrier interval behind group 0
  {
    int kt = kt_begin;
    for (; kt + 2 <= kt_end; kt += 2) {
      phase(I0{}, I0{}, kt); phase(I0{}, I1{}, kt); phase(I0{}, I2{}, kt); phase(I0{}, I3{}, kt);
      phase(I1{}, I0{}, kt + 1); phase(I1{}, I1{}, kt + 1); phase(I1{}, I2{}, kt + 1); phase(I1{}, I3{}, kt + 1);
    }
    if (kt < kt_end) { phase(I0{}, I0{}, kt); phase(I0{}, I1{}, kt); phase(I0{}, I2{}, kt); phase(I0{}, I3{}, kt); }
  }
  if (wr == 0) __builtin_amdgcn_s_barrier();
  __syncthreads();   // every fragment read and every DMA is done before the epilogue reuses the LDS
  write_tile<64, 64, 2, 2>(p, acc[0], wave_stage<64, 64>(smem, wave), lane, m0 + wr * 256 + wc * 64, n0);
  __syncthreads();
  write_tile<64, 64, 2, 2>(p, acc[1], wave_stage<64, 64>(smem, wave), lane, m0 + wr * 256 + wc * 64, n0 + 64);
}

#endif  // E4T_EXPERIMENTAL

#ifdef PP_TRACE
}  // namespace
extern "C" int e4t_debug_pp_trace(unsigned long long* out) {
  return (int)hipMemcpyFromSymbol(out, HIP_SYMBOL(g_pp_trace), sizeof(g_pp_trace));
}
namespace {
#endif

// ------------------------------------------------------------------------------------------------
// TN variant: C[M, N] = alpha * A^T . B with A = [K][M], B = [K][N] row-major bf16 — the contraction runs over the ROWS.
// This is the weight gradient dW = dY^T . X of every linear layer as the tensors lie in memory; the NT kernels above need
// both operands transposed first (two extra passes over dY and X per weight gradient).
//   * operand tiles are [64 k-rows][128 columns] (256-B rows), still moved by LDS-DMA, 4 rows per wave-instruction;
//   * the MFMA fragments (lane = output row / column, 8 consecutive k) are gathered with gfx950's transposing LDS read
//     ds_read_b64_tr_b16: within a 16-lane group, source lane i' supplies 4 contiguous elements and output lane i receives,
//     in slot j, element (i % 4) of source lane 4j + i / 4 (probed on hardware: tools/probe/tr_probe.hip).  Pointing
//     source lane i' at T[kbase + i'/4][n0 + 4 (i' % 4)] therefore hands lane i the column n0 + i of rows kbase .. kbase+3;
//     two such reads (rows +0 and +4) make one bf16x8 operand;
//   * bank conflicts: a 32-lane half reads 64 contiguous bytes from each of 4 consecutive rows; rows are 256 B apart = the
//     same banks, so 32-byte unit u of row r is stored at unit u ^ (2 (r & 3)) (swizzle applied on the DMA source side).
// 128 x 128 x 64 tile, 8 waves (wave tile 32 x 64), two LDS buffers, one barrier per K-tile, as the NT kernel.
// ------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(512) void gemm_tn_kernel(GemmArgs p) {
  constexpr int BM = 128, BN = 128, WGN = 2, WM = 32, WN = 64, FN = 2;
  constexpr int OPER = BK * 128;                 // elements per operand tile: 64 rows x 128 columns
  constexpr int TILE = 2 * OPER;
  __shared__ __attribute__((aligned(16))) bf16_t smem[2 * TILE];

  const int tid = threadIdx.x;
  const int lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wave / WGN, wn = wave % WGN;
  int tile_x, tile_y, sz = blockIdx.z;           // sz: split index (no batching in this variant)
  if (p.xcd3) { xcd_tile3(tile_x, tile_y, sz); p.zslab = sz; }
  else xcd_tile(tile_x, tile_y, p.group_m);
  const int m0 = tile_y * BM, n0 = tile_x * BN;

  const int nkt = (p.K + BK - 1) / BK;
  const int kt_begin = sz * p.ktiles_per_split;
  int kt_end = kt_begin + p.ktiles_per_split;
  if (kt_end > nkt) kt_end = nkt;

  // buffer-resource addressing as in the NT kernels: per-lane byte offset (row in piece, swizzled column chunk) fixed for the
  // whole loop, the wave-uniform scalar offset walks the contraction rows (+64 rows per K-tile); rows beyond K and columns
  // beyond M / N carry an out-of-range offset and read as zero
  const __amdgpu_buffer_rsrc_t rs_a = __builtin_amdgcn_make_buffer_rsrc((void*)p.A, 0, (int)p.a_bytes, 0x00020000);
  const __amdgpu_buffer_rsrc_t rs_b = __builtin_amdgcn_make_buffer_rsrc((void*)p.B, 0, (int)p.b_bytes, 0x00020000);
  constexpr unsigned OOB = 0xFFFF0000u;
  // DMA: piece q = 4 tile rows; wave w feeds pieces 2w, 2w+1 of each operand.  lane -> (row in piece, physical 16-B chunk)
  const int prow = lane >> 4, pchunk = lane & 15;
  const int lchunk = (((pchunk >> 1) ^ (2 * prow)) << 1) | (pchunk & 1);      // logical chunk stored at this physical slot
  const bool a_col_ok = m0 + lchunk * 8 < p.M, b_col_ok = n0 + lchunk * 8 < p.N;
  unsigned a_col[2], b_col[2], a_vo[2], b_vo[2];
#pragma unroll
  for (int i = 0; i < 2; ++i) {
    const int r = (wave * 2 + i) * 4 + prow;              // tile row of this lane in piece i
    a_col[i] = a_col_ok ? (unsigned)(((size_t)r * p.lda + m0 + lchunk * 8) * 2) : OOB;
    b_col[i] = b_col_ok ? (unsigned)(((size_t)r * p.ldb + n0 + lchunk * 8) * 2) : OOB;
  }
  int a_so = 0, b_so = 0;
  auto issue_tile = [&](int kt, bf16_t* buf) __attribute__((always_inline)) {
    if (kt == kt_begin || (kt + 1) * BK > p.K) {          // first tile, or the ragged last one: mask the rows beyond K
      a_so = kt * BK * p.lda * 2; b_so = kt * BK * p.ldb * 2;
#pragma unroll
      for (int i = 0; i < 2; ++i) {
        const bool row_ok = kt * BK + (wave * 2 + i) * 4 + prow < p.K;
        a_vo[i] = row_ok ? a_col[i] : OOB;
        b_vo[i] = row_ok ? b_col[i] : OOB;
      }
    } else {
      a_so += BK * p.lda * 2; b_so += BK * p.ldb * 2;
    }
#pragma unroll
    for (int i = 0; i < 2; ++i) buf_dma16(rs_a, a_vo[i], a_so, buf + (wave * 2 + i) * 512);
#pragma unroll
    for (int i = 0; i < 2; ++i) buf_dma16(rs_b, b_vo[i], b_so, buf + OPER + (wave * 2 + i) * 512);
  };

  f32x16 acc[1][FN];
#pragma unroll
  for (int j = 0; j < FN; ++j)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[0][j][r] = 0.f;

  // fragment gather addresses (elements) inside an operand tile for k-step 0, rows +0; +4 rows = +512, next k-step = +2048
  const int g16 = lane >> 4, il = lane & 15;
  const int frow = (g16 >> 1) * 8 + (il >> 2);
  auto frag_off = [&](int col0) {                 // col0: first of the 32 tile columns of the MFMA block
    const int col = col0 + 16 * (g16 & 1) + 4 * (il & 3);
    const int unit = (col >> 4) ^ (2 * (il >> 2));
    return frow * 128 + unit * 16 + (col & 15);
  };
  const int a_off = frag_off(wm * WM);
  int b_off[FN];
#pragma unroll
  for (int j = 0; j < FN; ++j) b_off[j] = OPER + frag_off(wn * WN + j * 32);

  if (kt_begin < kt_end) issue_tile(kt_begin, smem);
  auto body = [&](auto CURc, int kt) {
    constexpr int CUR = decltype(CURc)::value;
    wait_vmcnt<0>();
    loop_barrier();
    const bf16_t* st = smem + CUR * TILE;
    bf16x8 af[2], bfr[2][FN];
    af[0] = tr_frag(st + a_off, st + a_off + 512);
#pragma unroll
    for (int j = 0; j < FN; ++j) bfr[0][j] = tr_frag(st + b_off[j], st + b_off[j] + 512);
    if (kt + 1 < kt_end) issue_tile(kt + 1, smem + (CUR ^ 1) * TILE);
#pragma unroll
    for (int ks = 0; ks < BK / 16; ++ks) {
      const int c = ks & 1, n = c ^ 1;
      if (ks + 1 < BK / 16) {
        af[n] = tr_frag(st + a_off + (ks + 1) * 2048, st + a_off + (ks + 1) * 2048 + 512);
#pragma unroll
        for (int j = 0; j < FN; ++j) bfr[n][j] = tr_frag(st + b_off[j] + (ks + 1) * 2048, st + b_off[j] + (ks + 1) * 2048 + 512);
      }
#pragma unroll
      for (int j = 0; j < FN; ++j) acc[0][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[c], bfr[c][j], acc[0][j], 0, 0, 0);
    }
  };
  {
    int kt = kt_begin;
    for (; kt + 2 <= kt_end; kt += 2) {
      body(std::integral_constant<int, 0>{}, kt);
      body(std::integral_constant<int, 1>{}, kt + 1);
    }
    if (kt < kt_end) body(std::integral_constant<int, 0>{}, kt);
  }
  __syncthreads();
  write_tile<WM, WN, 1, FN>(p, acc, wave_stage<WM, WN>(smem, wave), lane, m0 + wm * WM, n0 + wn * WN);
}

// sums `nz` consecutive partial slabs starting at slab blockIdx.y*nz, then runs the epilogue for batch entry blockIdx.y
__global__ __launch_bounds__(256) void splitk_reduce_kernel(GemmArgs p, int nz) {
  const size_t total = (size_t)p.M * p.N;
  const int bz = blockIdx.y;
  if (!p.reduce_batch) {
    if (p.bias) p.bias += bz * p.strideBias;
    if (p.flags & E4T_OUT_F32) p.C = (float*)p.C + bz * p.strideC;
    else p.C = (bf16_t*)p.C + bz * p.strideC;
  }
  const float* ws = p.ws + (size_t)bz * nz * total;
  for (size_t idx = (size_t)blockIdx.x * 256 + threadIdx.x; idx < total; idx += (size_t)gridDim.x * 256) {
    float v = 0.f;
    for (int z = 0; z < nz; ++z) v += ws[(size_t)z * total + idx];
    const int row = (int)(idx / p.N), col = (int)(idx - (size_t)row * p.N);
    epilogue_store(p, v, row, col);
  }
}

// The same reduction for the common case — bf16 C, N and every leading dimension a multiple of 8, no accumulate — with 8
// columns per thread: the partial slabs are read as 2 x float4 per split with ALL splits of a trip in flight (the scalar
// version above walks them one 4-byte load at a time and answers the guarded bias / row-bias / residual loads of
// epilogue_store with one memory round trip each), bias / row bias / residual as 32- / 16-byte vectors, C as one 16-byte store.
// Same arithmetic, same summation order (z ascending), single rounding.
__global__ __launch_bounds__(256) void splitk_reduce8_kernel(GemmArgs p, int nz) {
  const int n8 = p.N >> 3;
  const size_t total = (size_t)p.M * p.N, total8 = (size_t)p.M * n8;
  const int bz = blockIdx.y;
  if (!p.reduce_batch) {
    if (p.bias) p.bias += bz * p.strideBias;
    p.C = (bf16_t*)p.C + bz * p.strideC;
  }
  const float* ws = p.ws + (size_t)bz * nz * total;
  const bf16_t* Rb = (const bf16_t*)p.residual;
  for (size_t i8 = (size_t)blockIdx.x * 256 + threadIdx.x; i8 < total8; i8 += (size_t)gridDim.x * 256) {
    const int row = (int)(i8 / n8), c8 = (int)(i8 - (size_t)row * n8) * 8;
    const size_t idx = (size_t)row * p.N + c8;
    float bi[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f}, rb[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f}, rs[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    if (p.bias) { const float4 a = *(const float4*)(p.bias + c8), b = *(const float4*)(p.bias + c8 + 4); bi[0] = a.x; bi[1] = a.y; bi[2] = a.z; bi[3] = a.w; bi[4] = b.x; bi[5] = b.y; bi[6] = b.z; bi[7] = b.w; }
    if (p.rowbias) {
      const float* q = p.rowbias + (size_t)(row / p.rows_per_batch) * p.ldrb + c8;
      const float4 a = *(const float4*)q, b = *(const float4*)(q + 4);
      rb[0] = a.x; rb[1] = a.y; rb[2] = a.z; rb[3] = a.w; rb[4] = b.x; rb[5] = b.y; rb[6] = b.z; rb[7] = b.w;
    }
    if (Rb) unpack8(*(const uint4*)(Rb + (size_t)row * p.ldr + c8), rs);
    float v[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    for (int z0 = 0; z0 < nz; z0 += 4) {
      float4 lo[4], hi[4];
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        const float* q = ws + (size_t)min(z0 + u, nz - 1) * total + idx;
        lo[u] = *(const float4*)q; hi[u] = *(const float4*)(q + 4);
      }
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        const float w = z0 + u < nz ? 1.f : 0.f;
        v[0] += lo[u].x * w; v[1] += lo[u].y * w; v[2] += lo[u].z * w; v[3] += lo[u].w * w;
        v[4] += hi[u].x * w; v[5] += hi[u].y * w; v[6] += hi[u].z * w; v[7] += hi[u].w * w;
      }
    }
#pragma unroll
    for (int k = 0; k < 8; ++k) {
      float x = v[k] * p.alpha + bi[k] + rb[k];
      if (p.flags & E4T_ACT_GELU) x = gelu_f(x);
      v[k] = x + rs[k];
    }
    *(uint4*)((bf16_t*)p.C + (size_t)row * p.ldc + c8) = pack8(v);
  }
}

// ... and for fp32 C without bias / row bias / residual / activation — every weight-gradient GEMM (gemm_tn_kernel, the batched head
// gradients): 4 columns per thread, float4 loads of all splits of a trip in flight, optional accumulate into C.  The scalar kernel
// above ran these at 1.2-3.4 TB/s (one 4-byte load per split and element: profiles/r04_prefetch_off_roofline_per_shape.csv, e.g.
// M320 N320 nz32 11.1 us, M16 N1280 nz129 35.8 us).  Same summation order (z ascending).
__global__ __launch_bounds__(256) void splitk_reduce4f_kernel(GemmArgs p, int nz) {
  const int n4 = p.N >> 2;
  const size_t total = (size_t)p.M * p.N, total4 = (size_t)p.M * n4;
  const int bz = blockIdx.y;
  float* Cf = (float*)p.C + (p.reduce_batch ? 0 : bz * p.strideC);
  const float* ws = p.ws + (size_t)bz * nz * total;
  for (size_t i4 = (size_t)blockIdx.x * 256 + threadIdx.x; i4 < total4; i4 += (size_t)gridDim.x * 256) {
    const int row = (int)(i4 / n4), c4 = (int)(i4 - (size_t)row * n4) * 4;
    const size_t idx = (size_t)row * p.N + c4;
    float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
    for (int z0 = 0; z0 < nz; z0 += 8) {
      float4 t[8];
#pragma unroll
      for (int u = 0; u < 8; ++u) t[u] = *(const float4*)(ws + (size_t)min(z0 + u, nz - 1) * total + idx);
#pragma unroll
      for (int u = 0; u < 8; ++u) {
        const float w = z0 + u < nz ? 1.f : 0.f;
        v.x += t[u].x * w; v.y += t[u].y * w; v.z += t[u].z * w; v.w += t[u].w * w;
      }
    }
    float4* c = (float4*)(Cf + (size_t)row * p.ldc + c4);
    float4 o = make_float4(v.x * p.alpha, v.y * p.alpha, v.z * p.alpha, v.w * p.alpha);
    if (p.flags & E4T_ACCUM) { const float4 a = *c; o.x += a.x; o.y += a.y; o.z += a.z; o.w += a.w; }
    *c = o;
  }
}

// What launch_gemm() will run for a problem: the ONE place where tiles and split-K are chosen (e4t_gemm_plan / e4t_conv3x3_plan
// export it, so that callers size workspaces and label timings from the launcher's own decision instead of mirroring it).
struct GemmPlan {
  int tile;        // 64 | 128 | 160 (128x160) | 256 (256x128 DMA) | 512 (256x256 ping-pong) | 640 (512x128 ping-pong) | 1128 / 1160 (persistent 256x128 / 256x160)
  int stages;      // LDS stages of the 64 / 128 / 160 DMA kernels
  bool kt32;       // 32-wide K-tiles (experimental codes 5064 / 5128 / 5256)
  bool general_epi;
  bool buf_ok;     // operands addressable through buffer resources (< 4 GB)
  int tm, tn, gx, gy;
  int splitk, ktiles_per_split;
  unsigned a_bytes, a2_bytes, b_bytes;
};

int device_cu_count() {
  static int ncu = [] {
    int dev = 0, n = 0;
    if (hipGetDevice(&dev) != hipSuccess || hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || n <= 0) n = 256;
    return n;
  }();
  return ncu;
}

// The persistent 256 x BN kernel (gemm_ps.hip) pays when its one-workgroup-per-CU rounds are (nearly) full.
bool ps_rounds_ok(long long units, int ncu) {
  if (units < ncu) return false;
  const long long rounds = (units + ncu - 1) / ncu;
  return units * 100 >= rounds * ncu * 85;
}

// ---- cost model of one launch of the 64 x 64 / 128 x 128 / 128 x 160 DMA tiles (round 4) ---------------------------------------------
// Below a few rounds of workgroups the time of these kernels is set by how many K-tiles a CU keeps in flight, not by the tile's FLOP per
// byte: per 64-wide K-tile a workgroup that is ALONE on its CU needs t1 (its own DMA latency / issue chain: 2 stages expose the whole
// L2 -> LDS latency, 3 - 4 stages hide it but cost the co-resident workgroups their LDS), several co-resident workgroups need tk each
// (the CU's L2 -> LDS bandwidth, ~60 GB/s per CU: 16 / 32 / 36 KiB per K-tile).  A launch costs
//     launch + rounds x (prologue + epilogue) + K-tiles per split x [full rounds x occ x tk + the last, partial round]  (+ the split-K reduce)
// with the workgroups dealt evenly to the CUs.  Constants (us) fitted by least squares to tools/sweep_small_m.py: 75 GEMM / conv shapes
// of the B = 1 / 4 / 16 steps x every (tile, stages, split) = 2900 graph-replayed, cold-operand timings, rms error 8.6 %; the
// argmin is within 6 % of the best measured variant on 72 of them (worst 1.18x), the rules it replaces were 5.8 % slower in total.
struct SmallGridPlan { int tile, stages, splitk; };
struct SmallGridCand { int tile, stages, tm, tn, occ; float t1, tk, epi; bool general_ok; };
SmallGridPlan small_grid_plan(int M, int N, int nkt, bool conv, bool general, int splitk_req, bool wants_colstats, int ncu) {
  static const SmallGridCand cands[] = {
      {64, 3, 64, 64, 3, 0.350f, 0.230f, 0.30f, true},    {64, 4, 64, 64, 2, 0.305f, 0.230f, 0.30f, false},
      {128, 2, 128, 128, 2, 0.875f, 0.544f, 1.08f, true}, {128, 4, 128, 128, 1, 0.596f, 0.544f, 1.08f, false},
      {160, 2, 128, 160, 2, 0.955f, 0.629f, 3.41f, false}, {160, 3, 128, 160, 1, 0.861f, 0.629f, 3.41f, false}};
  static const int splits[] = {1, 2, 3, 4, 6, 8, 12};
  const float launch = 1.6f, pro = 0.945f, red0 = 4.49f, red_bytes_per_us = 4.05e6f, conv_mul = 0.916f, stats_pass = 6.0f;
  SmallGridPlan best = {128, 2, 1};
  float best_t = 1e30f;
  for (const SmallGridCand& c : cands) {
    if (general && !c.general_ok) continue;        // (the GENERAL epilogue is instantiated for these two; the 128 x 160 one is register-bound)
    if (c.tile == 160 && N % 160 != 0) continue;
    for (int sk : splits) {
      if (splitk_req > 0 && sk != 1) continue;     // an explicit split: priced below with the request
      const int s = splitk_req > 0 ? (splitk_req < nkt ? splitk_req : nkt) : sk;
      if (s > 1 && splitk_req <= 0 && (nkt / s < 4 || (double)M * N * s * 4.0 > 256e6)) continue;
      const int kps = cdiv(nkt, s);
      const long long wgs = (long long)cdiv(M, c.tm) * cdiv(N, c.tn) * s;
      const long long per_cu = cdivl(wgs, ncu);
      const long long q = per_cu / c.occ, r = per_cu % c.occ;
      const float full = c.occ > 1 ? c.occ * c.tk : c.t1;
      const float rem = r == 0 ? 0.f : (r == 1 ? c.t1 : (float)r * c.tk);
      float per_kt = (float)q * full + rem;
      if (conv) per_kt *= conv_mul;
      const float rounds = (float)cdivl(per_cu, c.occ);
      float t = launch + rounds * (pro + c.epi) + (float)kps * per_kt;
      if (s > 1) {
        t += red0 + ((float)s * 4.f + 2.f) * (float)M * (float)N / red_bytes_per_us + rounds * c.epi * 0.5f;
        if (wants_colstats) t += stats_pass;       // a split launch leaves no column statistics: the consuming GroupNorm runs its own pass
      }
      if (t < best_t) { best_t = t; best = {c.tile, c.stages, s}; }
    }
  }
  return best;
}

GemmPlan plan_gemm(const GemmArgs& p, bool conv, int tile_hint, int splitk_req, int batch) {
  GemmPlan pl;
  const int nkt = cdiv(p.K, BK);
  // --- tile selection: fill >= ~1.5 waves of the 256 CUs with 128x128 tiles, else drop to 64x64 ---
  int stages = 2;                         // LDS stages of the 64 / 128 / 160 DMA kernels; a hint of 3128 / 4160 / ... forces 3 or 4
  int model_splitk = 0;                   // > 0: split-K chosen together with the tile by small_grid_plan
  bool kt32 = false;                      // 5128 / 5064 / 5256: the 32-wide K-tile variant (4 stages in the LDS of 2 x 64-wide ones; 5256: 3)
  if (tile_hint >= 3000 && tile_hint < 5000) { stages = tile_hint / 1000; tile_hint %= 1000; }
  else if (tile_hint >= 5000 && tile_hint < 6000) { kt32 = true; stages = 4; tile_hint %= 1000; }
  int tile = tile_hint;
#ifndef E4T_EXPERIMENTAL
  // The product library carries the tiles the planner chooses (64 / 128 / 160 with 2 - 4 LDS stages, 5256, 512, 2320) and nothing else:
  // the measured-and-rejected variants — 32-wide K-tiles on the 64 / 128 tiles, the 64-wide 256 x 128 tile, the 512 x 128 ping-pong
  // tile, the persistent streaming kernels of gemm_ps.hip — are built only with -DE4T_EXPERIMENTAL (csrc/build.sh:
  // E4T_EXPERIMENTAL=1).  A hint that names one of them gets the nearest product tile.
  if (kt32 && tile != 256) { kt32 = false; stages = 2; }
  if (tile != 64 && tile != 128 && tile != 160) stages = 2;
  if (tile == 256 && !kt32) { kt32 = true; stages = 3; }
  if (tile == 640) tile = 128;
  if (tile == 1128) tile = 128;
  if (tile == 1160) tile = 160;
#endif
  static const bool allow256 = getenv("E4T_GEMM_REGSTAGE") == nullptr;
#ifdef E4T_EXPERIMENTAL
  static const bool auto256 = getenv("E4T_GEMM_AUTO256") != nullptr;    // measured: 128x128/2-stage >= 256x128/3-stage on every E4T shape
#else
  const bool auto256 = false;
#endif
  const bool r2_rules = false;           // (round 2's rules: 112.5 vs 110.8 ms per step, profiles/r03_ab/r03c_*)
  const bool whole_k = p.K % BK == 0 && (!p.A2 || p.K1 % BK == 0);
  const bool ps_ok = allow256 && whole_k && batch == 1 && !p.reduce_batch && splitk_req <= 1;      // what gemm_ps_kernel accepts
  if (tile != 64 && tile != 128 && tile != 256 && tile != 160 && tile != 512 && tile != 640 && tile != 1128 && tile != 1160 && tile != 2320) {
    // measured on MI355X (tools/sweep_small.py): 128x128 wins from one full round of the 256 CUs, and already from
    // a quarter round when K is long (3x3 convs at the 16x16 / 8x8 levels) if split-K fills the chip
    const long long t256 = (long long)cdiv(p.M, 256) * cdiv(p.N, 128) * batch;
    const long long t128 = (long long)cdiv(p.M, 128) * cdiv(p.N, 128) * batch;
    tile = (allow256 && auto256 && t256 >= 256) ? 256 : (t128 >= 256 || (nkt >= 32 && t128 >= 64)) ? 128 : 64;
    // every channel count of the SD UNets is a multiple of 160 but 320 / 640 / 960 are not multiples of 128: a 128x160
    // tile has no N padding there (conv 640->640 @32x32: 863 vs 589 TF)
    if (allow256 && tile == 128 && p.N % 160 == 0 && p.N <= 960 && (long long)cdiv(p.M, 128) * (p.N / 160) * batch >= 128) tile = 160;
    // K-deep shapes whose N is a multiple of 256 and that fill the chip at least twice with 256x256 tiles (the VAE's 256-
    // and 512-channel convs): the ping-pong kernel (conv 512->512 @128^2: 1075 vs 928 TF, 8192^3: 1173 vs 971 TF)
    const bool no_pp = false;
    const long long tpp = (long long)cdiv(p.M, 256) * (p.N / 256) * batch;
    if (allow256 && !no_pp && tile == 128 && p.N % 256 == 0 && whole_k && nkt >= (conv ? 16 : (r2_rules ? 32 : 20)) && tpp >= 512) tile = 512;
    // ... and, with split-K, for the very K-deep shapes of the 16x16 level that give it less than one round of tiles (1280-channel
    // 3x3 convs, the GEGLU input gradient K = 10240): tools/sweep_step_shapes.py, conv 1280->1280 M4096 128 vs 141 us, conv
    // 1280->2560 209 vs 290 us, conv 2560->1280 221 vs 265 us, GEMM 4096x1280x10240 126 vs 138 us; at K = 5120 it loses.  (Round 3:
    // 256..511 tiles are NOT split any more — conv 1280->1280 M16384: 3 splits 545 us, 1 split 482 us, 128 x 160 tile 454 us.)
    if (allow256 && !no_pp && tile == 128 && p.N % 256 == 0 && whole_k && nkt >= 128 && tpp >= 64 && tpp < (r2_rules ? 512 : 256)) tile = 512;
    // one (nearly) full round of ping-pong tiles, K >= 1280: the ViT's qkv projection M4112 N3840 (255 tiles) 44.9 vs 49.5 us,
    // M4096 N3840 42.7 vs 48.3 us
    if (!r2_rules && allow256 && !no_pp && tile == 128 && p.N % 256 == 0 && whole_k && nkt >= 20 && tpp >= 224 && tpp <= 256) tile = 512;
    // The 512 x 128 variant of the same machine (tile code 640) is NOT chosen automatically: on the shapes it was built for
    // (the VAE's 128-channel convs, K = 1152 = 18 K-tiles) it measured 526 vs 588 TF/s for the 128 x 128 tile — one 160-KiB
    // workgroup per CU leaves nothing to overlap its (large) epilogue and prologue with, and 18 K-tiles do not amortise
    // them (tools/ab_pt.py).  E4T_GEMM_PT=1 turns the automatic choice on for experiments.
#ifdef E4T_EXPERIMENTAL
    static const bool auto_pt = getenv("E4T_GEMM_PT") != nullptr;
#else
    const bool auto_pt = false;
#endif
    if (auto_pt && allow256 && tile == 128 && p.N % 128 == 0 && p.N % 256 != 0 && p.K % BK == 0 && nkt >= 16 && !p.A2 &&
        (long long)cdiv(p.M, 512) * (p.N / 128) * batch >= 512) tile = 640;
    const bool general = (p.flags & E4T_ACT_GELU) || (p.rowbias && p.rows_per_batch % 32 != 0);
    // Round 3 (tools/sweep_ps.py on the step's own shapes, cold operands):
    //  * the 128 x 160 tile also for N > 960 when K is deep and the grid is large — conv 640->1920 M16384 350 vs 378 us, conv
    //    1280->1280 M16384 454 vs 482 (ping-pong) / 492 (128 x 128), GEMM M4096 N5120 K1280 60 vs 64 us; on short K it loses
    //    (M16384 N5120 K640: 145 vs 127 us), and its GELU instantiation is register-bound (M4112 N5120: 107 vs 85 us);
    if (!r2_rules && allow256 && tile == 128 && !general && p.N % 160 == 0 && p.N > 960 && nkt >= 20 &&
        (long long)cdiv(p.M, 128) * (p.N / 160) * batch >= 1024) tile = 160;
    //  * 256 x 128 with 32-wide K-tiles (8 waves, wave tile 64 x 64, three 24-KiB stages = two workgroups per CU; code 5256) for
    //    tall outputs whose N is a multiple of 128: half the B re-reads and 2/3 of the fragment LDS reads of the 128 x 128 tile per
    //    flop.  VAE conv 128->128 @512^2 1449 vs 1744 us, stride-2 conv 455 vs 493, conv_in GEMM M4194304 N128 K32 261 vs 386 (no
    //    padding of K to 64), GEMM M65536 N2560 K320 162 vs 185, M65536 N1280 K320 76 vs 81 us.
    if (!r2_rules && allow256 && tile == 128 && !general && (p.N % 128 == 0 || p.N < 128) &&
        (long long)cdiv(p.M, 256) * cdiv(p.N, 128) * batch >= 768) { tile = 256; kt32 = true; stages = 3; }
    //  * the 256 x 320 ping-pong tile (gemm_pq_kernel, code 2320) wherever N is a multiple of 320 and its rounds of one workgroup per
    //    CU come out full: 0.93 KB of LDS traffic per MFMA against 1.69 KB for the 128 x 160 tile.  Cold-operand sweep
    //    (profiles/r03_sweep_tiles_c.txt): conv 320->320 @64^2 105 vs 127 us (1154 TF/s), 640->640 @64^2 374 vs 458, 1280->1280
    //    M16384 354 vs 439 (1366 TF/s), nearest-x2 + conv 1280->1280 339 vs 472 (1424 TF/s); GEMM M65536 N320 K2560 103 vs 135,
    //    K1280 57 vs 73, K320 26 vs 28; M4096 N10240 K1280 100 vs 112 (ping-pong 256 x 256).  It loses below one round (M16384
    //    N640: 128 tiles) and on the K = 320 GEMMs wider than 320 (epilogue-bound: 83 vs 72 us at N1280).
    const bool no_pq = false;              // (without the 256 x 320 tile: 108.0 vs 106.4 ms per step, profiles/r03_ab/r03h_*)
    if (!r2_rules && !no_pq && allow256 && batch == 1 && (!general || !conv) && whole_k && p.N % 320 == 0) {
      const long long t320 = (long long)cdiv(p.M, 256) * (p.N / 320);
      const int ncu = device_cu_count();
      const double eff = (double)t320 / (double)(cdivl(t320, ncu) * ncu);
      if (t320 >= ncu && ((nkt >= 20 && eff >= 0.75) || (nkt >= 10 && eff >= 0.99) || (nkt >= 5 && p.N == 320 && eff >= 0.99))) {
        tile = 2320; kt32 = false; stages = 2;
      }
    }
    // The persistent 256 x 160 / 256 x 128 streaming kernel (gemm_ps.hip).  NOT chosen automatically: correct, but slower than the
    // tiles above on every shape of the step (its ping-pong phases are bound by the DMA-issue / fragment-read segment).
    // E4T_GEMM_PS=1 turns the automatic choice on for experiments.
#ifdef E4T_EXPERIMENTAL
    static const int auto_ps = getenv("E4T_GEMM_PS") ? atoi(getenv("E4T_GEMM_PS")) : E4T_GEMM_PS_DEFAULT;
#else
    const int auto_ps = 0;
#endif
    if (auto_ps && ps_ok && (tile == 128 || tile == 160)) {
      const int ncu = device_cu_count();
      const long long rows = cdiv(p.M, 256);
      if (p.N % 160 == 0 && ps_rounds_ok(rows * (p.N / 160), ncu)) tile = 1160;
      else if ((p.N % 128 == 0 || p.N < 128) && ps_rounds_ok(rows * cdiv(p.N, 128), ncu)) tile = 1128;
    }
    // Round 4: the 64 / 128 / 160 tiles, their LDS depth and split-K are arbitrated by a cost model of the launch (small_grid_plan above)
    // instead of the rules that chose among them until round 3; the rules above still decide WHETHER one of the big tiles runs.
    const bool no_model = false;           // (the round-3 rules instead of the cost model: C5 B = 1 44.4 vs 39.8 ms, profiles/r04_ab/r04s_*)
    if (!no_model && allow256 && (tile == 64 || tile == 128 || tile == 160) && batch == 1 && !p.reduce_batch && !p.panel_rows) {
      const SmallGridPlan sg = small_grid_plan(p.M, p.N, nkt, conv, general, splitk_req, p.colstats != nullptr, device_cu_count());
      tile = sg.tile; stages = sg.stages; model_splitk = sg.splitk;
    }
  }
  // epilogues with the exact GELU or a per-row row-bias lookup run the GENERAL instantiation, built for the 2-stage 64 / 128 / 160 tiles (and
  // the 3-stage 64 tile), the 256 x 256 / 256 x 320 ping-pong kernels and the persistent kernels
  const bool general_epi = (p.flags & E4T_ACT_GELU) || (p.rowbias && p.rows_per_batch % 32 != 0);
  if (general_epi) { if (!(tile == 64 && stages == 3)) stages = 2; kt32 = false; if (tile == 256 || tile == 640) tile = 128; }
  if (kt32 && tile != 128 && tile != 64 && tile != 256) kt32 = false;
  if (kt32 && tile == 256) stages = 3;
  if (tile == 256 && !allow256) tile = 128;
  if (tile == 160 && !allow256) tile = 128;
  if (tile == 640 && (!allow256 || p.A2)) tile = 128;
  if ((tile == 1128 || tile == 1160) && !ps_ok) tile = tile == 1160 && p.N % 160 == 0 ? 160 : 128;
  if (tile == 2320 && !allow256) tile = 128;
  if (tile == 2320 && general_epi && conv) tile = p.N % 160 == 0 ? 160 : 128;      // (the GENERAL instantiation of the 256 x 320 tile exists for GEMMs only)
  // row panels exist in the 256 x 320 ping-pong kernel only
  if (p.panel_rows && !(tile == 2320 && allow256)) tile = (p.N % 320 == 0 && whole_k && !conv) ? 2320 : tile;
  // The DMA kernels address their operands through buffer resources (32-bit byte offsets): operands beyond 4 GB fall back to
  // the register-staged kernel.  The ping-pong kernels additionally need whole K-tiles.
  bool buf_ok = true;
  {
    const unsigned long long lim = 0xFFFF0000ull;
    unsigned long long ab, a2b = 0, bb = ((unsigned long long)(p.N - 1) * p.ldb + p.K) * 2;
    if (conv) ab = (unsigned long long)((long long)p.M / ((long long)p.Hout * p.Wout)) * p.Hin * p.Win * p.Cin * 2;   // batch * Hin * Win * Cin
    else {
      const unsigned long long last = p.panel_rows ? (unsigned long long)((p.M - 1) / p.panel_rows) * p.panel_stride + p.panel_off + (p.M - 1) % p.panel_rows : (unsigned long long)(p.M - 1);
      ab = (last * p.lda + p.K1) * 2;
      if (p.A2) a2b = (last * p.lda2 + (p.K - p.K1)) * 2;
    }
    buf_ok = ab < lim && a2b < lim && bb < lim;
    if (tile == 512 && (!allow256 || !buf_ok || !whole_k)) tile = 128;
    if (tile == 2320 && (!buf_ok || !whole_k || p.N % 320 != 0)) tile = 128;      // (ragged N is only exercised for the narrower tiles)
    if (tile == 640 && (!buf_ok || p.K % BK != 0)) tile = 128;
    pl.a_bytes = (unsigned)(buf_ok ? ab : 0); pl.a2_bytes = (unsigned)(buf_ok ? a2b : 0); pl.b_bytes = (unsigned)(buf_ok ? bb : 0);
    if (!buf_ok && tile != 64) { tile = 128; kt32 = false; stages = 2; }      // the register-staged fallback exists as 128x128 and 64x64 only
  }
  // 256 = 256x128, 160 = 128x160, 512 = 256x256 ping-pong, 640 = 512x128 ping-pong, 1128 / 1160 = persistent 256x128 / 256x160
  const int tm = tile == 160 ? 128 : (tile == 512 || tile >= 1000 ? 256 : (tile == 640 ? 512 : tile));
  const int tn = tile == 256 ? 128 : (tile == 512 ? 256 : (tile == 640 ? 128 : (tile >= 1000 ? tile % 1000 : tile)));
  const bool pingpong = tile == 512 || tile == 2320;      // one 512-thread workgroup per CU
  const int gx = cdiv(p.N, tn), gy = cdiv(p.M, tm);
  // --- split-K: only when the grid underfills the chip and K is long ---
  int splitk = splitk_req;
  if (tile >= 1000 && tile < 2000) splitk = 1;
  if (p.panel_rows) splitk = 1;
  if (splitk <= 0 && model_splitk > 0 && (tile == 64 || tile == 128 || tile == 160)) splitk = model_splitk;
  if (splitk <= 0) {
    splitk = 1;
    const long long tiles = (long long)gx * gy * batch;
    if (pingpong && (tiles < 256 || !r2_rules)) {
      if (tiles < 256) {
        splitk = (int)(256 / tiles);                 // one 512-thread workgroup per CU: fill one round
        if (splitk > nkt / 16) splitk = nkt / 16;
      }                                              // (a second, partial round is not split: 3 splits of 320 tiles measured 545 vs 482 us)
    } else if (tile >= 128 && tiles < 512 && nkt >= 32) {
      // 2 workgroups/CU = 512 slots: aim at one full round (<= 256 tiles) or two (measured, tools/sweep_sk.py: 8x8 convs
      // 80 tiles -> 6 splits -16..19 %, 16x16 convs 320 tiles -> 3 splits -10..17 %), keeping >= 16 K-tiles per split
      splitk = (int)((tiles <= 256 ? 512 : 1024) / tiles);
      if (splitk > 8) splitk = 8;
      if (splitk > nkt / 16) splitk = nkt / 16;
      if (tiles > 256 && nkt < 128) splitk = 1;      // the second round only pays on long K (4096x1280x5120: 3 splits +4 %)
    } else if (tile == 64 && tiles < 256 && nkt >= 32) {
      splitk = (int)((512 + tiles - 1) / tiles);
      if (splitk > nkt / 16) splitk = nkt / 16;
    }
    if (splitk < 1) splitk = 1;
  }
  if (splitk > nkt) splitk = nkt;
  pl.ktiles_per_split = cdiv(nkt, splitk);
  pl.splitk = cdiv(nkt, pl.ktiles_per_split);
  pl.tile = tile; pl.stages = stages; pl.kt32 = kt32; pl.general_epi = general_epi; pl.buf_ok = buf_ok;
  pl.tm = tm; pl.tn = tn; pl.gx = gx; pl.gy = gy;
  return pl;
}

size_t plan_workspace_bytes(const GemmPlan& pl, const GemmArgs& p, int batch) {
  return (pl.splitk > 1 || p.reduce_batch) ? (size_t)pl.splitk * batch * p.M * p.N * sizeof(float) : 0;
}

// Tail rows (gemm_common.h, gemm_tail): a dense GEMM whose M is a multiple of 128 plus at most 32 rows — the CLIP-ViT's 16 x 257 =
// 4112 token rows — is planned for its first M - r rows; the r tail rows are computed at the end of the same launch.  Returns the plan
// and sets `tail` (0: the ordinary plan over all M rows).  The plan of the M - r rows must be a single pass of one of the LDS-DMA kernels
// that carry the tail code (128 x 160, 256 x 256 and 256 x 320 tiles): anything else falls back to the plan over all rows.
GemmPlan plan_gemm_tail(const GemmArgs& p, bool conv, int tile_hint, int splitk_req, int batch, int& tail) {
  const bool no_tail = false;              // (without the tail stage: README recipe 134.1 vs 133.5 ms, C4 139.6 vs 139.4, headline step equal; profiles/r05_ab)
  static const bool use_dma = getenv("E4T_GEMM_REGSTAGE") == nullptr;
  tail = 0;
  const int r = p.M % 128;
  if (!no_tail && use_dma && !conv && batch == 1 && !p.reduce_batch && !p.A2 && !p.panel_rows && !p.colstats && splitk_req <= 1 && r > 0 && r <= 32 &&
      p.M - r >= 1024 && p.K % 16 == 0 && p.lda % 8 == 0 && p.ldb % 8 == 0 && (((uintptr_t)p.A | (uintptr_t)p.B) & 15) == 0) {
    GemmArgs q = p;
    q.M = p.M - r;
    const GemmPlan pl = plan_gemm(q, conv, tile_hint, splitk_req, batch);
    const bool kernel_ok = pl.buf_ok && pl.splitk == 1 && (pl.tile == 160 || pl.tile == 512 || pl.tile == 2320);      // the kernels that carry gemm_tail()
    if (kernel_ok) { tail = r; return pl; }
  }
  return plan_gemm(p, conv, tile_hint, splitk_req, batch);
}

// the float4 reduction: fp32 C, nothing in the epilogue but alpha (and accumulate), 16-byte aligned rows
bool reduce4f_ok(const GemmArgs& p) {
  return p.ws && (p.flags & E4T_OUT_F32) && !(p.flags & E4T_ACT_GELU) && !p.bias && !p.rowbias && !p.residual && p.N % 4 == 0 && p.ldc % 4 == 0 &&
         p.strideC % 4 == 0 && (((uintptr_t)p.C | (uintptr_t)p.ws) & 15) == 0;
}

int launch_gemm(GemmArgs p, bool conv, int tile_hint, size_t ws_bytes, int splitk_req, int batch, hipStream_t st) {
  int tail = 0;
  const GemmPlan pl = plan_gemm_tail(p, conv, tile_hint, splitk_req, batch, tail);
  const int m_all = p.M;                 // (launch log: the caller's shape)
  if (tail) { p.M -= tail; p.tail_row0 = p.M; p.tail_rows = tail; }      // the tile grid covers [0, M - tail); gemm_tail() the rest
  const int nkt = cdiv(p.K, BK);
  const int tile = pl.tile, stages = pl.stages, gx = pl.gx, gy = pl.gy;
  const bool kt32 = pl.kt32, general_epi = pl.general_epi, buf_ok = pl.buf_ok;
  int splitk = pl.splitk;
  p.ktiles_per_split = pl.ktiles_per_split;
  p.a_bytes = pl.a_bytes; p.a2_bytes = pl.a2_bytes; p.b_bytes = pl.b_bytes;
  // stride-1 3x3 convs walk K channel-chunk-major (gemm_common.h, cm_step) in every DMA kernel
  const bool tap_major = false;            // (tap-major K order: 109.7 vs 107.5 ms per step, 2.4-5.2 x the HBM traffic; profiles/r03_ab/r03p_*)
  p.chan_major = conv && !tap_major && buf_ok && p.mode == E4T_CONV_S1 && p.Cin % BK == 0 && p.K == 9 * p.Cin && batch == 1 &&
                 (unsigned long long)p.a_bytes + (unsigned long long)(p.Win + 1) * p.Cin * 2 < 0xFFFF0000ull;
  static const bool allow256 = getenv("E4T_GEMM_REGSTAGE") == nullptr;
  (void)allow256;
  const bool need_ws = splitk > 1 || p.reduce_batch;
  const size_t need = (size_t)splitk * batch * p.M * p.N * sizeof(float);
  if (need_ws && (p.ws == nullptr || ws_bytes < need)) {
    if (splitk_req > 1 || p.reduce_batch)
      E4T_FAIL(-12, "gemm: split-K=%d batch=%d needs %zu workspace bytes, have %zu", splitk, batch, need, ws_bytes);
    splitk = 1;  // auto mode: fall back to a single pass rather than fail
    p.ktiles_per_split = nkt;
  }
  if (!(splitk > 1 || p.reduce_batch)) p.ws = nullptr;
  p.splitk = splitk;
  p.group_m = 8;                           // row panels per raster group (swept in round 4: profiles/r04_ab/r04f_*)
  p.fast_epi = !(p.flags & (E4T_OUT_F32 | E4T_ACCUM | E4T_RES_F32)) && p.N % 8 == 0 && p.ldc % 8 == 0 && ((uintptr_t)p.C & 15) == 0 &&
               (p.strideC % 8 == 0) && (!p.residual || (p.ldr % 8 == 0 && ((uintptr_t)p.residual & 15) == 0));
  p.fast_f32 = (p.flags & E4T_OUT_F32) && !(p.flags & E4T_ACCUM) && !p.rowbias && (!p.residual || (p.flags & E4T_RES_F32)) &&
               (!(p.flags & E4T_ACT_GELU) || general_epi);
  if (p.colstats && (p.ws || !p.fast_epi || p.M % 32 != 0 || batch != 1)) p.colstats = nullptr;   // only the bf16 single-pass epilogue produces them
  const int stats_written = p.colstats != nullptr;
  // split-K / batch reduction: the 8-columns-per-thread kernel when C is bf16 and everything it touches is 16-byte aligned
  const bool vec8 = p.ws && !(p.flags & (E4T_OUT_F32 | E4T_ACCUM | E4T_RES_F32)) && p.N % 8 == 0 && p.ldc % 8 == 0 && ((uintptr_t)p.C & 15) == 0 &&
                    p.strideC % 8 == 0 && (!p.residual || (p.ldr % 8 == 0 && ((uintptr_t)p.residual & 15) == 0)) &&
                    (!p.bias || (((uintptr_t)p.bias & 15) == 0 && p.strideBias % 4 == 0)) &&
                    (!p.rowbias || (((uintptr_t)p.rowbias & 15) == 0 && p.ldrb % 4 == 0)) && ((uintptr_t)p.ws & 15) == 0;
  dim3 grid(gx, gy, splitk * batch), block(256);
  static const bool use_dma = getenv("E4T_GEMM_REGSTAGE") == nullptr;   // A/B switch: register-staged reference kernel
  if (e4t_launch_log_enabled()) {
    // algorithmic bytes: every operand element once (conv: the input map once, not once per tap), the output once
    const double osz = (p.flags & E4T_OUT_F32) ? 4.0 : 2.0;
    const double a_el = conv ? (double)(p.M / ((long long)p.Hout * p.Wout)) * p.Hin * p.Win * p.Cin : (double)p.M * p.K * batch;
    double by = 2.0 * a_el + 2.0 * (double)p.N * p.K * (p.strideB || batch == 1 ? batch : 1) + osz * (double)p.M * p.N * (p.reduce_batch ? 1 : batch);
    if (p.residual) by += ((p.flags & E4T_RES_F32) ? 4.0 : 2.0) * (double)p.M * p.N;
    if (p.flags & E4T_ACCUM) by += osz * (double)p.M * p.N;
    const char* sym = !(use_dma && buf_ok) ? "gemm_kernel"
                      : tile == 512 && conv && conv_pps_ok(p, batch) ? (general_epi ? "gemm_pps_kernel<true>" : "gemm_pps_kernel<false>")
                      : tile == 512 ? (general_epi ? (conv ? (p.chan_major ? "gemm_pp_kernel<1, true, true>" : "gemm_pp_kernel<1, true, false>") : "gemm_pp_kernel<0, true, false>")
                                                   : (conv ? (p.chan_major ? "gemm_pp_kernel<1, false, true>" : "gemm_pp_kernel<1, false, false>") : "gemm_pp_kernel<0, false, false>"))
                      : tile == 640 ? (conv ? "gemm_pt_kernel<1>" : "gemm_pt_kernel<0>")
                      : tile == 256 && !kt32 ? (conv ? "gemm_dma_kernel<256, 128, 4, 2, 1, 3, false, 64>" : "gemm_dma_kernel<256, 128, 4, 2, 0, 3, false, 64>")
                      : tile == 256 ? (conv ? (conv_strip_ok(p, splitk, batch) ? "conv_strip_kernel<2>" : "gemm_dma_kernel<256, 128, 4, 2, 1, 3, false, 32>")
                                            : "gemm_dma_kernel<256, 128, 4, 2, 0, 3, false, 32>")
                      : nullptr;
    char symbuf[64];
    if (!sym && tile >= 2000) {
      snprintf(symbuf, sizeof(symbuf), "gemm_pq_kernel<%d, %d, %s>", conv ? 1 : 0, tile - 2000, general_epi ? "true" : "false");
      sym = symbuf;
    }
    if (!sym && tile >= 1000) {
      snprintf(symbuf, sizeof(symbuf), "gemm_ps_kernel<%d, %d, %s>", conv ? 1 : 0, tile - 1000, general_epi ? "true" : "false");
      sym = symbuf;
    }
    if (!sym) {
      snprintf(symbuf, sizeof(symbuf), "gemm_dma_kernel<%d, %d, %d, %d, %d, %d, %s, %d>", tile == 64 ? 64 : 128, tile == 160 ? 160 : tile, tile == 64 ? 2 : 4,
               tile == 160 ? 1 : 2, conv ? 1 : 0, stages, general_epi ? "true" : "false", kt32 ? 32 : 64);      // as rocprofv3 prints the symbol
      sym = symbuf;
    }
    if (conv) E4T_LOG_LAUNCH("%s|conv mode%d %dx%d->%dx%d Cin%d Cout%d M%d splitk%d|%.0f|%.0f", sym, p.mode, p.Hin, p.Win, p.Hout, p.Wout, p.Cin, p.N,
                             p.M, splitk, by, 2.0 * p.M * p.N * (double)p.K);
    else E4T_LOG_LAUNCH("%s|gemm M%d N%d K%d batch%d splitk%d flags%d|%.0f|%.0f", sym, m_all, p.N, p.K, batch, splitk, p.flags,
                        by + (double)tail * (2.0 * p.K + (osz + (p.residual ? ((p.flags & E4T_RES_F32) ? 4.0 : 2.0) : 0.0)) * p.N), 2.0 * m_all * p.N * (double)p.K * batch);
    if (p.ws) E4T_LOG_LAUNCH("%s|M%d N%d nz%d|%.0f|0", vec8 ? "splitk_reduce8_kernel" : reduce4f_ok(p) ? "splitk_reduce4f_kernel" : "splitk_reduce_kernel",
                             p.M, p.N, p.reduce_batch ? splitk * batch : splitk, 4.0 * (double)p.M * p.N * splitk * batch + osz * (double)p.M * p.N);
  }
  if (p.panel_rows && !(use_dma && buf_ok && tile == 2320 && splitk == 1 && !p.ws))
    E4T_FAIL(-22, "gemm: row panels need the 256 x 320 ping-pong tile (N %% 320 == 0, K %% 64 == 0, no split-K); the plan chose tile %d", tile);
  if (use_dma && buf_ok) {
    if (tile == 2320) {
      block = dim3(512);
      if (conv) hipLaunchKernelGGL((gemm_pq_kernel<1, 320, false>), grid, block, 0, st, p);
      else if (general_epi) hipLaunchKernelGGL((gemm_pq_kernel<0, 320, true>), grid, block, 0, st, p);
      else hipLaunchKernelGGL((gemm_pq_kernel<0, 320, false>), grid, block, 0, st, p);
#ifdef E4T_EXPERIMENTAL
    } else if (tile >= 1000) {
      static const bool ps_pre = getenv("E4T_PS_PRE") == nullptr || atoi(getenv("E4T_PS_PRE")) != 0;      // A/B switch
      p.ps_pre = ps_pre && p.fast_epi && nkt >= 2 && (!p.rowbias || p.rows_per_batch % 256 == 0);
      const int rc = e4t_launch_gemm_ps(&p, conv ? 1 : 0, tile - 1000, general_epi ? 1 : 0, device_cu_count(), st);
      if (rc < 0) return rc;
#endif
    } else if (tile == 512) {
      block = dim3(512);
      if (conv && conv_pps_ok(p, batch)) {
        if (general_epi) hipLaunchKernelGGL((gemm_pps_kernel<true>), grid, block, 0, st, p);
        else hipLaunchKernelGGL((gemm_pps_kernel<false>), grid, block, 0, st, p);
      } else if (general_epi) {
        if (conv && p.chan_major) hipLaunchKernelGGL((gemm_pp_kernel<1, true, true>), grid, block, 0, st, p);
        else if (conv) hipLaunchKernelGGL((gemm_pp_kernel<1, true>), grid, block, 0, st, p);
        else hipLaunchKernelGGL((gemm_pp_kernel<0, true>), grid, block, 0, st, p);
      } else {
#ifdef E4T_EXPERIMENTAL
        // measured and rejected (round 6): the 16-wave 256 x 256 strip kernel EQUALS the ping-pong kernel on every conv shape of the step (1072 / 1007 /
        // 526 / 254 / 126 us against 1046-1082 / 995-1011 / 515-531 / 253 / 125-128) although it fills a third of the A bytes — with 64 x 64 wave tiles
        // it needs one fragment ds_read per MFMA, 125 B/clk/CU of LDS reads at the MFMA peak against the LDS's 128
        static const bool strip256 = getenv("E4T_CONV_STRIP256") != nullptr && atoi(getenv("E4T_CONV_STRIP256")) != 0;
        if (conv && strip256 && conv_strip_ok(p, splitk, batch)) hipLaunchKernelGGL(conv_strip_kernel<4>, grid, dim3(1024), 0, st, p);
        else
#endif
        if (conv && p.chan_major) hipLaunchKernelGGL((gemm_pp_kernel<1, false, true>), grid, block, 0, st, p);
        else if (conv) hipLaunchKernelGGL((gemm_pp_kernel<1>), grid, block, 0, st, p);
        else hipLaunchKernelGGL((gemm_pp_kernel<0>), grid, block, 0, st, p);
      }
#ifdef E4T_EXPERIMENTAL
    } else if (tile == 640) {
      block = dim3(512);
      if (conv) hipLaunchKernelGGL((gemm_pt_kernel<1>), grid, block, 0, st, p);
      else hipLaunchKernelGGL((gemm_pt_kernel<0>), grid, block, 0, st, p);
#endif
    } else if (tile == 256 && kt32) {
      block = dim3(512);       // experimental (5256): 256 x 128 with 32-wide K-tiles, 3 x 24 KiB stages = two workgroups per CU
      if (conv && conv_strip_ok(p, splitk, batch)) hipLaunchKernelGGL(conv_strip_kernel<2>, grid, block, 0, st, p);
      else if (conv) hipLaunchKernelGGL((gemm_dma_kernel<256, 128, 4, 2, 1, 3, false, 32>), grid, block, 0, st, p);
      else hipLaunchKernelGGL((gemm_dma_kernel<256, 128, 4, 2, 0, 3, false, 32>), grid, block, 0, st, p);
#ifdef E4T_EXPERIMENTAL
    } else if (tile == 256) {
      block = dim3(512);
      if (conv) hipLaunchKernelGGL((gemm_dma_kernel<256, 128, 4, 2, 1, 3>), grid, block, 0, st, p);
      else hipLaunchKernelGGL((gemm_dma_kernel<256, 128, 4, 2, 0, 3>), grid, block, 0, st, p);
#endif
    } else {
      // 64 / 128 / 160 tiles: 2 LDS stages and 2 workgroups per CU by default; 3 or 4 stages (one workgroup per CU, 2-3 K-tiles
      // in flight) when the grid cannot give a CU two workgroups anyway — see the stage choice above
#define E4T_LAUNCH_DMA_STAGES(BM_, BN_, WGM_, WGN_)                                                                       \
    if (!general_epi && stages == 4) { if (conv) hipLaunchKernelGGL((gemm_dma_kernel<BM_, BN_, WGM_, WGN_, 1, 4>), grid, block, 0, st, p); \
                       else hipLaunchKernelGGL((gemm_dma_kernel<BM_, BN_, WGM_, WGN_, 0, 4>), grid, block, 0, st, p); }    \
    else if (!general_epi && stages == 3) { if (conv) hipLaunchKernelGGL((gemm_dma_kernel<BM_, BN_, WGM_, WGN_, 1, 3>), grid, block, 0, st, p); \
                            else hipLaunchKernelGGL((gemm_dma_kernel<BM_, BN_, WGM_, WGN_, 0, 3>), grid, block, 0, st, p); } \
    else
#define E4T_LAUNCH_DMA(BM_, BN_, WGM_, WGN_, NT_)                                                                         \
  do {                                                                                                                   \
    block = dim3(NT_);                                                                                                   \
    E4T_LAUNCH_DMA_STAGES(BM_, BN_, WGM_, WGN_)                                                                          \
    if (general_epi) { if (conv) hipLaunchKernelGGL((gemm_dma_kernel<BM_, BN_, WGM_, WGN_, 1, 2, true>), grid, block, 0, st, p); \
                            else hipLaunchKernelGGL((gemm_dma_kernel<BM_, BN_, WGM_, WGN_, 0, 2, true>), grid, block, 0, st, p); } \
    else { if (conv) hipLaunchKernelGGL((gemm_dma_kernel<BM_, BN_, WGM_, WGN_, 1, 2>), grid, block, 0, st, p);            \
           else hipLaunchKernelGGL((gemm_dma_kernel<BM_, BN_, WGM_, WGN_, 0, 2>), grid, block, 0, st, p); }               \
  } while (0)
      // 128x128: 8 waves (wave tile 32x64): ~4 waves/SIMD at 2 workgroups/CU hide the DMA/LDS latency that the 4-wave
      // version of the same tile exposed (measured +5..18 % on every E4T shape, 8192^3: 956 -> 980 TF)
#ifdef E4T_EXPERIMENTAL
      if (kt32 && tile == 128) {
        block = dim3(512);
        if (conv) hipLaunchKernelGGL((gemm_dma_kernel<128, 128, 4, 2, 1, 4, false, 32>), grid, block, 0, st, p);
        else hipLaunchKernelGGL((gemm_dma_kernel<128, 128, 4, 2, 0, 4, false, 32>), grid, block, 0, st, p);
      } else if (kt32 && tile == 64) {
        block = dim3(256);
        if (conv) hipLaunchKernelGGL((gemm_dma_kernel<64, 64, 2, 2, 1, 4, false, 32>), grid, block, 0, st, p);
        else hipLaunchKernelGGL((gemm_dma_kernel<64, 64, 2, 2, 0, 4, false, 32>), grid, block, 0, st, p);
      } else
#endif
      if (tile == 64 && general_epi && stages == 3) {
        block = dim3(256);
        if (conv) hipLaunchKernelGGL((gemm_dma_kernel<64, 64, 2, 2, 1, 3, true>), grid, block, 0, st, p);
        else hipLaunchKernelGGL((gemm_dma_kernel<64, 64, 2, 2, 0, 3, true>), grid, block, 0, st, p);
      } else if (tile == 160) E4T_LAUNCH_DMA(128, 160, 4, 1, 256);
      else if (tile == 128) E4T_LAUNCH_DMA(128, 128, 4, 2, 512);
      else E4T_LAUNCH_DMA(64, 64, 2, 2, 256);
#undef E4T_LAUNCH_DMA
#undef E4T_LAUNCH_DMA_STAGES
    }
  } else if (tile == 128) {
    if (conv) hipLaunchKernelGGL((gemm_kernel<128, 128, 2, 2, 1>), grid, block, 0, st, p);
    else hipLaunchKernelGGL((gemm_kernel<128, 128, 2, 2, 0>), grid, block, 0, st, p);
  } else {
    if (conv) hipLaunchKernelGGL((gemm_kernel<64, 64, 2, 2, 1>), grid, block, 0, st, p);
    else hipLaunchKernelGGL((gemm_kernel<64, 64, 2, 2, 0>), grid, block, 0, st, p);
  }
  E4T_CHECK_LAUNCH("gemm_kernel");
  if (p.ws) {
    const size_t total = (size_t)p.M * p.N;
    int blocks = (int)((total + 255) / 256);
    if (blocks > 2048) blocks = 2048;
    const int nz = p.reduce_batch ? splitk * batch : splitk;
    if (vec8) {
      int b8 = (int)((total / 8 + 255) / 256);
      if (b8 > 2048) b8 = 2048;
      hipLaunchKernelGGL(splitk_reduce8_kernel, dim3(b8, p.reduce_batch ? 1 : batch), dim3(256), 0, st, p, nz);
    } else if (reduce4f_ok(p)) {
      int b4 = (int)((total / 4 + 255) / 256);
      if (b4 > 2048) b4 = 2048;
      hipLaunchKernelGGL(splitk_reduce4f_kernel, dim3(b4, p.reduce_batch ? 1 : batch), dim3(256), 0, st, p, nz);
    } else {
      hipLaunchKernelGGL(splitk_reduce_kernel, dim3(blocks, p.reduce_batch ? 1 : batch), dim3(256), 0, st, p, nz);
    }
    E4T_CHECK_LAUNCH("splitk_reduce_kernel");
  }
  return (use_dma && buf_ok) ? stats_written : 0;
}

}  // namespace

namespace {

void fill_gemm_args(const e4t_gemm_desc* d, GemmArgs& p) {
  memset(&p, 0, sizeof(p));
  p.zslab = -1;
  p.A = (const bf16_t*)d->A; p.A2 = (const bf16_t*)d->A2; p.K1 = d->A2 ? d->K1 : d->K; p.lda = d->lda; p.lda2 = d->lda2;
  p.B = (const bf16_t*)d->B; p.ldb = d->ldb; p.C = d->C; p.ldc = d->ldc; p.bias = d->bias; p.residual = d->residual; p.ldr = d->ldr;
  p.rowbias = d->rowbias; p.rows_per_batch = d->rows_per_batch > 0 ? d->rows_per_batch : 1; p.ldrb = d->ldrb > 0 ? d->ldrb : d->N;
  p.M = d->M; p.N = d->N; p.K = d->K; p.alpha = d->alpha; p.flags = d->flags; p.ws = (float*)d->workspace;
  p.strideA = d->strideA; p.strideB = d->strideB; p.strideC = d->strideC; p.strideBias = d->strideBias;
  p.reduce_batch = (d->flags & E4T_REDUCE_BATCH) ? 1 : 0;
  p.colstats = d->colstats;
  p.panel_rows = d->panel_rows; p.panel_stride = d->panel_stride; p.panel_off = d->panel_off;
}

void fill_conv_args(const e4t_conv_desc* d, GemmArgs& p) {
  const int Cin = d->Cin, Cout = d->Cout;
  memset(&p, 0, sizeof(p));
  p.zslab = -1;
  p.A = (const bf16_t*)d->X; p.K1 = 9 * Cin; p.Hin = d->Hin; p.Win = d->Win; p.Cin = Cin; p.Hout = d->Hout; p.Wout = d->Wout;
  p.mode = d->mode; p.B = (const bf16_t*)d->W; p.ldb = 9 * Cin; p.C = d->Y; p.ldc = Cout; p.bias = d->bias;
  p.residual = d->residual; p.ldr = Cout; p.rowbias = d->rowbias; p.rows_per_batch = d->Hout * d->Wout; p.ldrb = d->ldrb > 0 ? d->ldrb : Cout;
  p.M = d->B * d->Hout * d->Wout; p.N = Cout; p.K = 9 * Cin; p.alpha = 1.f; p.flags = d->flags; p.ws = (float*)d->workspace;
  p.colstats = d->colstats;
}

// split count of the TN (weight-gradient) kernel: few output tiles, very long K -> fill the 512 workgroup slots by splitting K
int tn_splitk(int M, int N, int K, int req) {
  const int nkt = cdiv(K, BK);
  const long long tiles = (long long)cdiv(N, 128) * cdiv(M, 128);
  int splitk = req;
  if (splitk <= 0) {
    splitk = (int)(512 / tiles);
    if (splitk > 32) splitk = 32;
    if (splitk > nkt / 16) splitk = nkt / 16;
    if (splitk < 1) splitk = 1;
  }
  if (splitk > nkt) splitk = nkt;
  return cdiv(nkt, cdiv(nkt, splitk));
}

void export_plan(const GemmPlan& pl, const GemmArgs& p, int batch, e4t_gemm_plan_t* out, int tail = 0) {
  out->tile = pl.kt32 ? 5000 + pl.tile : pl.tile; out->tile_m = pl.tm; out->tile_n = pl.tn; out->splitk = pl.splitk;
  out->workspace_bytes = plan_workspace_bytes(pl, p, batch);
  out->tail_rows = tail; out->stages = (!pl.kt32 && (pl.tile == 64 || pl.tile == 128 || pl.tile == 160)) ? pl.stages : 0;
}

}  // namespace

extern "C" int e4t_gemm_plan(const e4t_gemm_desc* d, e4t_gemm_plan_t* out) {
  E4T_REQUIRE(d && out && d->M > 0 && d->N > 0 && d->K > 0, "gemm_plan: bad arguments");
  GemmArgs p;
  fill_gemm_args(d, p);
  const int batch = d->batch > 0 ? d->batch : 1;
  int tail = 0;
  const GemmPlan pl = plan_gemm_tail(p, false, d->tile, d->splitk, batch, tail);
  export_plan(pl, p, batch, out, tail);      // (no workspace with tail rows: their plan is a single pass)
  return 0;
}

extern "C" int e4t_conv3x3_plan(const e4t_conv_desc* d, e4t_gemm_plan_t* out) {
  E4T_REQUIRE(d && out && d->B > 0 && d->Hout > 0 && d->Wout > 0 && d->Cin > 0 && d->Cout > 0, "conv3x3_plan: bad arguments");
  GemmArgs p;
  fill_conv_args(d, p);
  export_plan(plan_gemm(p, true, d->tile, d->splitk, 1), p, 1, out);
  return 0;
}

extern "C" int e4t_gemm_tn_plan(const e4t_gemm_desc* d, e4t_gemm_plan_t* out) {
  E4T_REQUIRE(d && out && d->M > 0 && d->N > 0 && d->K > 0, "gemm_tn_plan: bad arguments");
  out->tile = 128; out->tile_m = out->tile_n = 128; out->tail_rows = out->stages = 0;
  out->splitk = tn_splitk(d->M, d->N, d->K, d->splitk);
  out->workspace_bytes = out->splitk > 1 ? (size_t)out->splitk * d->M * d->N * sizeof(float) : 0;
  return 0;
}

extern "C" int e4t_gemm_nt(const e4t_gemm_desc* d, e4t_stream stream) {
  E4T_REQUIRE(d && d->A && d->B && d->C, "gemm_nt: null operand");
  E4T_REQUIRE(d->M > 0 && d->N > 0 && d->K > 0, "gemm_nt: bad shape M=%d N=%d K=%d", d->M, d->N, d->K);
  E4T_REQUIRE(d->K % 8 == 0 && d->lda % 8 == 0 && d->ldb % 8 == 0, "gemm_nt: K, lda, ldb must be multiples of 8 (16-B rows)");
  E4T_REQUIRE(((uintptr_t)d->A & 15) == 0 && ((uintptr_t)d->B & 15) == 0, "gemm_nt: A/B must be 16-B aligned");
  if (d->A2) {
    E4T_REQUIRE(d->K1 % BK == 0 && d->K1 > 0 && d->K1 < d->K && d->lda2 % 8 == 0 && ((uintptr_t)d->A2 & 15) == 0,
                "gemm_nt: two-source A needs K1 %% 64 == 0, 0 < K1 < K");
  }
  E4T_REQUIRE(!d->rowbias || d->rows_per_batch > 0, "gemm_nt: rowbias needs rows_per_batch");
  const int batch = d->batch > 0 ? d->batch : 1;
  E4T_REQUIRE(batch == 1 || (d->strideA % 8 == 0 && d->strideB % 8 == 0), "gemm_nt: batch strides must keep 16-B alignment");
  E4T_REQUIRE(!(d->flags & E4T_REDUCE_BATCH) || !d->A2, "gemm_nt: reduce-batch with two-source A unsupported");
  if (d->panel_rows) {
    E4T_REQUIRE(d->panel_rows > 0 && d->panel_rows % 256 == 0 && d->panel_stride >= d->panel_rows && d->panel_off >= 0 && d->M % d->panel_rows == 0,
                "gemm_nt: row panels need panel_rows %% 256 == 0, panel_stride >= panel_rows, M %% panel_rows == 0");
    E4T_REQUIRE(batch == 1 && !d->rowbias && !d->colstats && !d->A2 && !(d->flags & (E4T_ACCUM | E4T_REDUCE_BATCH)) && d->splitk <= 1,
                "gemm_nt: row panels do not combine with batch / row bias / column statistics / two-source A / accumulate / split-K");
  }
  GemmArgs p;
  fill_gemm_args(d, p);
  return launch_gemm(p, false, d->tile, d->workspace_bytes, d->splitk, batch, (hipStream_t)stream);
}

extern "C" int e4t_gemm_tn(const e4t_gemm_desc* d, e4t_stream stream) {
  E4T_REQUIRE(d && d->A && d->B && d->C, "gemm_tn: null operand");
  E4T_REQUIRE(d->M > 0 && d->N > 0 && d->K > 0 && d->batch <= 1 && !d->A2 && !d->rowbias, "gemm_tn: bad / unsupported arguments");
  E4T_REQUIRE(!(d->flags & E4T_ACT_GELU), "gemm_tn: the GELU epilogue is not built for the TN kernel");
  E4T_REQUIRE(d->M % 8 == 0 && d->N % 8 == 0 && d->lda % 8 == 0 && d->ldb % 8 == 0 && ((uintptr_t)d->A & 15) == 0 && ((uintptr_t)d->B & 15) == 0,
              "gemm_tn: M, N, lda, ldb must be multiples of 8 and the operands 16-byte aligned");
  GemmArgs p;
  memset(&p, 0, sizeof(p));
  p.zslab = -1;
  p.A = (const bf16_t*)d->A; p.lda = d->lda; p.B = (const bf16_t*)d->B; p.ldb = d->ldb;
  p.C = d->C; p.ldc = d->ldc; p.bias = d->bias; p.residual = d->residual; p.ldr = d->ldr;
  p.M = d->M; p.N = d->N; p.K = d->K; p.K1 = d->K; p.alpha = d->alpha; p.flags = d->flags & ~E4T_REDUCE_BATCH;
  p.rows_per_batch = 1; p.ldrb = d->N;
  p.ws = (float*)d->workspace;
  const int nkt = cdiv(p.K, BK), gx = cdiv(p.N, 128), gy = cdiv(p.M, 128);
  {   // operand extents for the buffer resources (32-bit byte offsets)
    const unsigned long long ab = ((unsigned long long)(p.K - 1) * p.lda + p.M) * 2, bb = ((unsigned long long)(p.K - 1) * p.ldb + p.N) * 2;
    E4T_REQUIRE(ab < 0xFFFF0000ull && bb < 0xFFFF0000ull, "gemm_tn: operands beyond 4 GB are not supported");
    p.a_bytes = (unsigned)ab; p.b_bytes = (unsigned)bb;
  }
  int splitk = tn_splitk(p.M, p.N, p.K, d->splitk);
  p.ktiles_per_split = cdiv(nkt, splitk);
  const size_t need = (size_t)splitk * p.M * p.N * sizeof(float);
  if (splitk > 1 && (p.ws == nullptr || d->workspace_bytes < need)) {
    if (d->splitk > 1) E4T_FAIL(-12, "gemm_tn: split-K=%d needs %zu workspace bytes, have %zu", splitk, need, d->workspace_bytes);
    splitk = 1;
    p.ktiles_per_split = nkt;
  }
  if (splitk <= 1) p.ws = nullptr;
  p.splitk = splitk;
  p.group_m = 8;
  p.xcd3 = splitk > 1;                     // (2-D raster: gemm_tn M960 N320 K65536 90.6 vs 67.4 us)
  p.fast_epi = !(p.flags & (E4T_OUT_F32 | E4T_ACCUM | E4T_RES_F32)) && p.N % 8 == 0 && p.ldc % 8 == 0 && ((uintptr_t)p.C & 15) == 0 &&
               (!p.residual || (p.ldr % 8 == 0 && ((uintptr_t)p.residual & 15) == 0));
  hipStream_t st = (hipStream_t)stream;
  if (e4t_launch_log_enabled()) {
    const double osz = (p.flags & E4T_OUT_F32) ? 4.0 : 2.0;
    E4T_LOG_LAUNCH("gemm_tn_kernel|gemm_tn M%d N%d K%d splitk%d flags%d|%.0f|%.0f", p.M, p.N, p.K, splitk, p.flags,
                   2.0 * (double)p.K * (p.M + p.N) + osz * (double)p.M * p.N * ((p.flags & E4T_ACCUM) ? 2 : 1), 2.0 * p.M * p.N * (double)p.K);
    if (p.ws) E4T_LOG_LAUNCH("%s|M%d N%d nz%d|%.0f|0", reduce4f_ok(p) ? "splitk_reduce4f_kernel" : "splitk_reduce_kernel", p.M, p.N, splitk,
                             4.0 * (double)p.M * p.N * splitk + osz * (double)p.M * p.N);
  }
  hipLaunchKernelGGL(gemm_tn_kernel, dim3(gx, gy, splitk), dim3(512), 0, st, p);
  E4T_CHECK_LAUNCH("gemm_tn_kernel");
  if (p.ws) {
    const size_t total = (size_t)p.M * p.N;
    int blocks = (int)((total + 255) / 256);
    if (blocks > 2048) blocks = 2048;
    if (reduce4f_ok(p)) {
      int b4 = (int)((total / 4 + 255) / 256);
      if (b4 > 2048) b4 = 2048;
      hipLaunchKernelGGL(splitk_reduce4f_kernel, dim3(b4, 1), dim3(256), 0, st, p, splitk);
    } else {
      hipLaunchKernelGGL(splitk_reduce_kernel, dim3(blocks, 1), dim3(256), 0, st, p, splitk);
    }
    E4T_CHECK_LAUNCH("splitk_reduce_kernel");
  }
  return 0;
}

extern "C" int e4t_conv3x3(const e4t_conv_desc* d, e4t_stream stream) {
  E4T_REQUIRE(d && d->X && d->W && d->Y, "conv3x3: null operand");
  const int Cin = d->Cin, Cout = d->Cout, Hin = d->Hin, Win = d->Win, Hout = d->Hout, Wout = d->Wout, mode = d->mode;
  E4T_REQUIRE(Cin % BK == 0, "conv3x3: Cin=%d must be a multiple of 64 (pad the input channels)", Cin);
  E4T_REQUIRE(mode >= E4T_CONV_S1 && mode <= E4T_CONV_S2A, "conv3x3: bad mode %d", mode);
  E4T_REQUIRE(d->B > 0 && Hin > 0 && Win > 0 && Hout > 0 && Wout > 0 && Cout > 0, "conv3x3: bad geometry");
  if (mode == E4T_CONV_S1) E4T_REQUIRE(Hout == Hin && Wout == Win, "conv3x3 S1: output must equal input size");
  if (mode == E4T_CONV_S2) E4T_REQUIRE(Hout == (Hin - 1) / 2 + 1 && Wout == (Win - 1) / 2 + 1, "conv3x3 S2: bad output size");
  if (mode == E4T_CONV_UP2) E4T_REQUIRE(Hout == 2 * Hin && Wout == 2 * Win, "conv3x3 UP2: output must be 2x input");
  if (mode == E4T_CONV_S2T) E4T_REQUIRE(Hin == (Hout - 1) / 2 + 1 && Win == (Wout - 1) / 2 + 1, "conv3x3 S2T: bad sizes");
  if (mode == E4T_CONV_S2A) E4T_REQUIRE(Hout == (Hin - 2) / 2 + 1 && Wout == (Win - 2) / 2 + 1, "conv3x3 S2A: bad output size");
  GemmArgs p;
  fill_conv_args(d, p);
  return launch_gemm(p, true, d->tile, d->workspace_bytes, d->splitk, 1, (hipStream_t)stream);
}
