// GroupNorm(+SiLU) and LayerNorm, forward and backward, on NHWC bf16 activations (gfx950).
//
// All of these are HBM-bound streaming kernels: every thread owns fixed channel chunks and walks
// pixels, so global accesses are 8/16-byte vectors with consecutive lanes on consecutive addresses;
// statistics are fp32.  GroupNorm accepts the input split over two sources along C — that is how
// torch.cat([hidden, skip], dim=1) in the up-blocks (e4t/models/unet_2d_blocks.py:1795,1883) is
// consumed without a concat copy — and its backward writes dX back into two destinations.
//
// Reference call sites: [3P diffusers] ResnetBlock2D norm1/norm2 + SiLU (constructed at
// e4t/models/unet_2d_blocks.py:481,760,881,1732,1855), Transformer2DModel.norm (eps 1e-6,
// e4t/models/transformer_2d.py:149,253), conv_norm_out (e4t/models/unet_2d_condition.py:275-278,
// 554-556), nn.LayerNorm in BasicTransformerBlock (e4t/models/attention.py:259,268,273) and in the
// open_clip ViT (e4t/encoder.py:154).
#include "common.h"
#include <stdlib.h>
#include "../../include/e4t_hip.h"

namespace {

// Thread mapping shared by all GroupNorm kernels: a pixel row of C channels is QW = C/8 chunks of 16 bytes.
//   QW <= 256: the block's 256 threads form PG = 256/QW pixel groups x QW chunk lanes (every thread busy);
//   QW  > 256: one pixel group, each thread owns chunks tid and tid+256 (C <= 4096).
constexpr int GN_MAXS = 2;
// pixels a thread has in flight per loop trip: all GN_U x (1-3) 16-byte loads of a trip are issued before the first is consumed.
// One load per trip left ~16 KiB in flight per CU = 2 TB/s on the mid-size maps (Little's law at ~2 us); 4 reach the HBM rate.
constexpr int GN_U = 4;          // (one-slot kernels; the two-slot ones, C > 2048, carry twice the state and use GN_U / 2)

struct GNSrc {
  const bf16_t* x1; const bf16_t* x2; int C1, C2;
};
__device__ __forceinline__ uint4 gn_load8(const GNSrc& s, size_t pix, int c) {
  if (c < s.C1) return *(const uint4*)(s.x1 + pix * s.C1 + c);
  return *(const uint4*)(s.x2 + pix * s.C2 + (c - s.C1));
}
struct GNMap {
  int QW, PG, pg, nslot;
  int ch[GN_MAXS];      // chunk index per slot, -1 = inactive
  __device__ GNMap(int C) {
    QW = C >> 3;
    const int tid = threadIdx.x;
    if (QW <= 256) {
      PG = 256 / QW; nslot = 1; pg = tid / QW;
      ch[0] = (tid < PG * QW) ? tid - pg * QW : -1; ch[1] = -1;
      if (ch[0] < 0) pg = 0;
    } else {
      PG = 1; nslot = 2; pg = 0;
      ch[0] = tid; ch[1] = (tid + 256 < QW) ? tid + 256 : -1;
    }
  }
};

// Reduce per-thread per-channel pairs (a[s][j], b[s][j]) over the block's pixel groups, then over each group's
// channels.  lds: [PG][C][2] scratch followed by [C][2] channel totals.  Fixed order -> deterministic.
__device__ __forceinline__ void gn_block_reduce(const GNMap& m, int C, int G, float (*a)[8], float (*b)[8], float* lds,
                                                 const float* gamma_or_null, float* group_out, float* chan_out) {
  float* chs = lds + (size_t)m.PG * C * 2;
#pragma unroll
  for (int s = 0; s < GN_MAXS; ++s)
    if (m.ch[s] >= 0)
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        const int c = m.ch[s] * 8 + j;
        lds[((size_t)m.pg * C + c) * 2] = a[s][j];
        lds[((size_t)m.pg * C + c) * 2 + 1] = b[s][j];
      }
  __syncthreads();
  for (int c = threadIdx.x; c < C; c += 256) {
    float ta = 0.f, tb = 0.f;
    for (int g = 0; g < m.PG; ++g) { ta += lds[((size_t)g * C + c) * 2]; tb += lds[((size_t)g * C + c) * 2 + 1]; }
    if (chan_out) { chan_out[c * 2] = ta; chan_out[c * 2 + 1] = tb; }
    if (gamma_or_null) { ta *= gamma_or_null[c]; tb *= gamma_or_null[c]; }
    chs[c * 2] = ta; chs[c * 2 + 1] = tb;
  }
  __syncthreads();
  if ((int)threadIdx.x < G) {
    const int cpg = C / G;
    float ta = 0.f, tb = 0.f;
    for (int c = threadIdx.x * cpg; c < (threadIdx.x + 1) * cpg; ++c) { ta += chs[c * 2]; tb += chs[c * 2 + 1]; }
    group_out[threadIdx.x * 2] = ta; group_out[threadIdx.x * 2 + 1] = tb;
  }
}

// ---- straight-line inner loops ---------------------------------------------------------------------------------------------
// Every GroupNorm kernel walks its chunk GN_U pixels per trip.  The loads of a trip must sit in ONE basic block: the first
// version guarded each load (inactive thread / pixel past the chunk end) and the compiler answered every guarded load with
// its own s_waitcnt vmcnt(0) — one memory round trip per 16 bytes per thread, 2 TB/s on the 64x64-level maps whatever the
// occupancy; the per-channel gamma / beta / mean / rstd fetches of the prologue were 16-32 such round trips in a row before
// the first pixel was touched (ISA inspection, round 2).  Now: the slot count NS is a template parameter, inactive threads
// alias chunk 0 and out-of-range pixels alias the chunk's last pixel (loads unconditional, only stores / accumulations are
// masked), the two-source select is a pointer select, and the per-channel constants come from 32-byte vector loads plus an
// LDS copy of the per-group statistics.
__device__ __forceinline__ uint4 gn_load8u(const GNSrc& s, size_t pix, int c) {
  const bf16_t* p = c < s.C1 ? s.x1 + pix * s.C1 + c : s.x2 + pix * s.C2 + (c - s.C1);
  return *(const uint4*)p;
}
__device__ __forceinline__ void ld8f(const float* p, float* o) {
  const float4 a = *(const float4*)p, b = *(const float4*)(p + 4);
  o[0] = a.x; o[1] = a.y; o[2] = a.z; o[3] = a.w; o[4] = b.x; o[5] = b.y; o[6] = b.z; o[7] = b.w;
}
template <int NS>
struct GNSlots {
  int c[NS];       // first channel of the slot's 8-channel chunk (chunk 0 for an inactive slot)
  bool act[NS];
  __device__ GNSlots(const GNMap& m) {
#pragma unroll
    for (int i = 0; i < NS; ++i) { act[i] = m.ch[i] >= 0; c[i] = act[i] ? m.ch[i] * 8 : 0; }
  }
};
// the (b, g) statistics of this batch entry in LDS: mr[g*2] = mean, mr[g*2+1] = rstd   (G <= 256)
__device__ __forceinline__ void gn_stage_stats(float* mr, const float* mean_rstd_b, int G) {
  for (int i = threadIdx.x; i < G * 2; i += 256) mr[i] = mean_rstd_b[i];
  __syncthreads();
}

// partial[b][chunk][g][2] = (sum, sumsq) of x over this chunk's pixels and group g's channels
template <int NS>
__device__ __forceinline__ void gn_stats_body(const GNSrc& s, const GNMap& m, int b, int HW, int p0, int p1, float (*sm)[8], float (*sq)[8]) {
  constexpr int U = NS == 1 ? GN_U : GN_U / 2;
  const GNSlots<NS> sl(m);
  for (int p = p0 + m.pg; p < p1; p += U * m.PG) {
    uint4 vx[U][NS];
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const int pp = min(p + u * m.PG, p1 - 1);
#pragma unroll
      for (int i = 0; i < NS; ++i) vx[u][i] = gn_load8u(s, (size_t)b * HW + pp, sl.c[i]);
    }
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const bool in = p + u * m.PG < p1;
#pragma unroll
      for (int i = 0; i < NS; ++i) {
        float f[8];
        unpack8(vx[u][i], f);
        const float w = (in && sl.act[i]) ? 1.f : 0.f;
#pragma unroll
        for (int j = 0; j < 8; ++j) { const float v = f[j] * w; sm[i][j] += v; sq[i][j] += v * v; }
      }
    }
  }
}
template <int NS>
__global__ __launch_bounds__(256) void gn_stats_kernel(GNSrc s, int HW, int G, int pix_per_chunk, float* partial) {
  extern __shared__ float lds[];
  const int C = s.C1 + s.C2;
  const GNMap m(C);
  const int b = blockIdx.y, chunk = blockIdx.x;
  const int p0 = chunk * pix_per_chunk;
  int p1 = p0 + pix_per_chunk; if (p1 > HW) p1 = HW;
  float sm[GN_MAXS][8], sq[GN_MAXS][8];
#pragma unroll
  for (int i = 0; i < GN_MAXS; ++i)
#pragma unroll
    for (int j = 0; j < 8; ++j) sm[i][j] = sq[i][j] = 0.f;
  if (p0 < p1) gn_stats_body<NS>(s, m, b, HW, p0, p1, sm, sq);
  gn_block_reduce(m, C, G, sm, sq, lds, nullptr, partial + ((size_t)b * gridDim.x + chunk) * G * 2, nullptr);
}

// mean_rstd[b][g] = (mean, rstd).  One wave per (b, g): lane c sums chunks c, c+64, ... in double, then a fixed-order
// butterfly (a single thread walking the 64 chunk partials was a 13-us chain of dependent loads, 110 times per step).
__device__ __forceinline__ double wave_sum_f64(double v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  return v;
}
__global__ __launch_bounds__(256) void gn_finalize_kernel(const float* partial, int nchunk, int G, int BG, float inv_n, float eps,
                                                          float* mean_rstd) {
  const int i = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
  if (i >= BG) return;
  const int b = i / G, g = i - b * G;
  double s = 0.0, q = 0.0;
  for (int c = lane; c < nchunk; c += 64) {
    const float2 pp = *(const float2*)(partial + (((size_t)b * nchunk + c) * G + g) * 2);
    s += pp.x; q += pp.y;
  }
  s = wave_sum_f64(s); q = wave_sum_f64(q);
  if (lane == 0) {
    const double mean = s * inv_n;
    double var = q * inv_n - mean * mean;
    if (var < 0.0) var = 0.0;
    mean_rstd[i * 2] = (float)mean;
    mean_rstd[i * 2 + 1] = (float)(1.0 / sqrt(var + (double)eps));
  }
}

// Block-cooperative finalize (fused into the consumers: saves one launch per GroupNorm call each way): every wave walks
// groups wave, wave+4, ...; lane c sums chunks c, c+64, ... in double, fixed-order butterfly — the same order as the
// stand-alone gn_finalize_kernel, so both paths are bitwise identical; fin(g, sum0, sum1) runs on lane 0 of the wave.
template <typename Fin>
__device__ __forceinline__ void gn_block_sum_partials(const float* partial_b /* [nchunk][G][2] of this batch */, int nchunk, int G, Fin fin) {
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63, nw = blockDim.x >> 6;
  for (int g = wave; g < G; g += nw) {
    double s = 0.0, q = 0.0;
    for (int c = lane; c < nchunk; c += 64) {
      const float2 pp = *(const float2*)(partial_b + ((size_t)c * G + g) * 2);
      s += pp.x; q += pp.y;
    }
    s = wave_sum_f64(s); q = wave_sum_f64(q);
    if (lane == 0) fin(g, s, q);
  }
  __syncthreads();
}

// y = act(gamma * (x - mean) * rstd + beta), written as one contiguous (B*HW, C) bf16 matrix
// FUSED: mean / rstd are finalised from the stats kernel's chunk partials in the prologue (and written to mean_rstd_out by
// chunk 0 of every batch entry for the backward) instead of by a separate finalize launch.
template <int NS, bool SILU>
__device__ __forceinline__ void gn_apply_body(const GNSrc& s, const GNMap& m, int b, int HW, int C, int cpg, int p0, int p1, const float* mr,
                                              const float* gamma, const float* beta, bf16_t* y, int silu) {
  constexpr int U = NS == 1 ? GN_U : GN_U / 2;
  const GNSlots<NS> sl(m);
  float sc[NS][8], sh[NS][8];
#pragma unroll
  for (int i = 0; i < NS; ++i) {
    float ga[8], be[8];
    ld8f(gamma + sl.c[i], ga);
    ld8f(beta + sl.c[i], be);
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      const int g = (sl.c[i] + j) / cpg;
      sc[i][j] = mr[g * 2 + 1] * ga[j];
      sh[i][j] = be[j] - mr[g * 2] * sc[i][j];
    }
  }
  for (int p = p0 + m.pg; p < p1; p += U * m.PG) {
    uint4 vx[U][NS];
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const int pp = min(p + u * m.PG, p1 - 1);
#pragma unroll
      for (int i = 0; i < NS; ++i) vx[u][i] = gn_load8u(s, (size_t)b * HW + pp, sl.c[i]);
    }
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const bool in = p + u * m.PG < p1;
      const size_t pix = (size_t)b * HW + p + u * m.PG;
#pragma unroll
      for (int i = 0; i < NS; ++i) {
        float f[8];
        unpack8(vx[u][i], f);
#pragma unroll
        for (int j = 0; j < 8; ++j) {
          const float z = f[j] * sc[i][j] + sh[i][j];
          f[j] = SILU ? silu_f(z) : z;
        }
        if (in && sl.act[i]) *(uint4*)(y + pix * C + sl.c[i]) = pack8(f);
      }
    }
  }
}
template <bool FUSED, int NS>
__global__ __launch_bounds__(256) void gn_apply_kernel(GNSrc s, const float* mean_rstd, const float* gamma, const float* beta,
                                                       bf16_t* y, int HW, int G, int pix_per_chunk, int silu,
                                                       const float* partial, int npart, float inv_n, float eps, float* mean_rstd_out) {
  const int C = s.C1 + s.C2, cpg = C / G;
  const GNMap m(C);
  const int b = blockIdx.y;
  __shared__ float mr_lds[512];
  if (FUSED) {
    gn_block_sum_partials(partial + (size_t)b * npart * G * 2, npart, G, [&](int g, double sm, double sq) {
      const double mean = sm * inv_n;
      double var = sq * inv_n - mean * mean;
      if (var < 0.0) var = 0.0;
      const float mf = (float)mean, rf = (float)(1.0 / sqrt(var + (double)eps));
      mr_lds[g * 2] = mf; mr_lds[g * 2 + 1] = rf;
      if (blockIdx.x == 0) { mean_rstd_out[((size_t)b * G + g) * 2] = mf; mean_rstd_out[((size_t)b * G + g) * 2 + 1] = rf; }
    });
  } else {
    gn_stage_stats(mr_lds, mean_rstd + (size_t)b * G * 2, G);
  }
  const int p0 = blockIdx.x * pix_per_chunk;
  int p1 = p0 + pix_per_chunk; if (p1 > HW) p1 = HW;
  if (p0 >= p1) return;
  if (silu) gn_apply_body<NS, true>(s, m, b, HW, C, cpg, p0, p1, mr_lds, gamma, beta, y, silu);
  else gn_apply_body<NS, false>(s, m, b, HW, C, cpg, p0, p1, mr_lds, gamma, beta, y, silu);
}

// Backward pass 1.  With z = gamma*xhat + beta, dz = dy * act'(z):
//   partial[b][chunk][g] = ( sum_c gamma_c * sum_p dz , sum_c gamma_c * sum_p dz*xhat )
//   chan_partial[b][chunk][c] = ( sum_p dz , sum_p dz*xhat )      (optional: gives dbeta, dgamma)
template <int NS, bool SILU>
__device__ __forceinline__ void gn_bwd_stats_body(const GNSrc& s, const GNMap& m, int b, int HW, int C, int cpg, int p0, int p1, const float* mr,
                                                  const bf16_t* dy, const float* gamma, const float* beta, int silu, float (*s1)[8], float (*s2)[8]) {
  constexpr int U = NS == 1 ? GN_U : GN_U / 2;
  const GNSlots<NS> sl(m);
  float mu[NS][8], rs[NS][8], ga[NS][8], be[NS][8];
#pragma unroll
  for (int i = 0; i < NS; ++i) {
    ld8f(gamma + sl.c[i], ga[i]);
    ld8f(beta + sl.c[i], be[i]);
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      const int g = (sl.c[i] + j) / cpg;
      mu[i][j] = mr[g * 2]; rs[i][j] = mr[g * 2 + 1];
    }
  }
  for (int p = p0 + m.pg; p < p1; p += U * m.PG) {
    uint4 vx[U][NS], vd[U][NS];
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const size_t pix = (size_t)b * HW + min(p + u * m.PG, p1 - 1);
#pragma unroll
      for (int i = 0; i < NS; ++i) {
        vx[u][i] = gn_load8u(s, pix, sl.c[i]);
        vd[u][i] = *(const uint4*)(dy + pix * C + sl.c[i]);
      }
    }
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const bool in = p + u * m.PG < p1;
#pragma unroll
      for (int i = 0; i < NS; ++i) {
        float f[8], d[8];
        unpack8(vx[u][i], f);
        unpack8(vd[u][i], d);
        const float w = (in && sl.act[i]) ? 1.f : 0.f;
#pragma unroll
        for (int j = 0; j < 8; ++j) {
          const float xh = (f[j] - mu[i][j]) * rs[i][j];
          float dz = d[j] * w;
          if (SILU) dz *= dsilu_f(ga[i][j] * xh + be[i][j]);
          s1[i][j] += dz; s2[i][j] += dz * xh;
        }
      }
    }
  }
}
template <int NS>
__global__ __launch_bounds__(256) void gn_bwd_stats_kernel(GNSrc s, const bf16_t* dy, const float* mean_rstd, const float* gamma,
                                                           const float* beta, int HW, int G, int pix_per_chunk, int silu,
                                                           float* partial, float* chan_partial) {
  extern __shared__ float lds[];
  __shared__ float mr_lds[512];
  const int C = s.C1 + s.C2, cpg = C / G;
  const GNMap m(C);
  const int b = blockIdx.y, chunk = blockIdx.x;
  const int p0 = chunk * pix_per_chunk;
  int p1 = p0 + pix_per_chunk; if (p1 > HW) p1 = HW;
  gn_stage_stats(mr_lds, mean_rstd + (size_t)b * G * 2, G);
  float s1[GN_MAXS][8], s2[GN_MAXS][8];
#pragma unroll
  for (int i = 0; i < GN_MAXS; ++i)
#pragma unroll
    for (int j = 0; j < 8; ++j) s1[i][j] = s2[i][j] = 0.f;
  if (p0 < p1) {
    if (silu) gn_bwd_stats_body<NS, true>(s, m, b, HW, C, cpg, p0, p1, mr_lds, dy, gamma, beta, silu, s1, s2);
    else gn_bwd_stats_body<NS, false>(s, m, b, HW, C, cpg, p0, p1, mr_lds, dy, gamma, beta, silu, s1, s2);
  }
  gn_block_reduce(m, C, G, s1, s2, lds, gamma, partial + ((size_t)b * gridDim.x + chunk) * G * 2,
                  chan_partial ? chan_partial + ((size_t)b * gridDim.x + chunk) * C * 2 : nullptr);
}

// Backward pass 2: dx = rstd * (dz*gamma - (S1 + xhat*S2)/n) (+ add), split into dx1 | dx2 along C
// gsum == nullptr: (S1, S2) are reduced from bwd_stats' chunk partials in the prologue (no finalize launch)
template <int NS, bool SILU>
__device__ __forceinline__ void gn_bwd_apply_body(const GNSrc& s, const GNMap& m, int b, int HW, int C, int cpg, int p0, int p1, const float* mr,
                                                  const float* gs, float inv_n, const bf16_t* dy, const float* gamma, const float* beta,
                                                  const bf16_t* add1, const bf16_t* add2, bf16_t* dx1, bf16_t* dx2, int silu) {
  constexpr int U = NS == 1 ? GN_U : GN_U / 2;
  const GNSlots<NS> sl(m);
  float mu[NS][8], rs[NS][8], ga[NS][8], be[NS][8], g1[NS][8], g2[NS][8];
#pragma unroll
  for (int i = 0; i < NS; ++i) {
    ld8f(gamma + sl.c[i], ga[i]);
    ld8f(beta + sl.c[i], be[i]);
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      const int g = (sl.c[i] + j) / cpg;
      mu[i][j] = mr[g * 2]; rs[i][j] = mr[g * 2 + 1];
      g1[i][j] = gs[g * 2] * inv_n; g2[i][j] = gs[g * 2 + 1] * inv_n;
    }
  }
  // the gradient arriving through the block's shortcut: a zero page stands in when a source has none, so that the load stays unconditional
  const bool has1 = add1 != nullptr, has2 = add2 != nullptr;
  for (int p = p0 + m.pg; p < p1; p += U * m.PG) {
    uint4 vx[U][NS], vd[U][NS], va[U][NS];
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const size_t pix = (size_t)b * HW + min(p + u * m.PG, p1 - 1);
#pragma unroll
      for (int i = 0; i < NS; ++i) {
        const int c = sl.c[i];
        vx[u][i] = gn_load8u(s, pix, c);
        vd[u][i] = *(const uint4*)(dy + pix * C + c);
        const bool first = c < s.C1;
        const bf16_t* ap = first ? (has1 ? add1 + pix * s.C1 + c : dy + pix * C + c) : (has2 ? add2 + pix * s.C2 + (c - s.C1) : dy + pix * C + c);
        va[u][i] = *(const uint4*)ap;            // (a duplicate of the dy load, ignored below, when that source has no shortcut gradient)
      }
    }
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const bool in = p + u * m.PG < p1;
      const size_t pix = (size_t)b * HW + p + u * m.PG;
#pragma unroll
      for (int i = 0; i < NS; ++i) {
        const int c = sl.c[i];
        const bool first = c < s.C1;
        const float wa = (first ? has1 : has2) ? 1.f : 0.f;
        float f[8], d[8], a[8];
        unpack8(vx[u][i], f);
        unpack8(vd[u][i], d);
        unpack8(va[u][i], a);
#pragma unroll
        for (int j = 0; j < 8; ++j) {
          const float xh = (f[j] - mu[i][j]) * rs[i][j];
          float dz = d[j];
          if (SILU) dz *= dsilu_f(ga[i][j] * xh + be[i][j]);
          f[j] = rs[i][j] * (dz * ga[i][j] - g1[i][j] - xh * g2[i][j]) + a[j] * wa;
        }
        if (in && sl.act[i]) {
          if (first) *(uint4*)(dx1 + pix * s.C1 + c) = pack8(f);
          else *(uint4*)(dx2 + pix * s.C2 + (c - s.C1)) = pack8(f);
        }
      }
    }
  }
}
template <int NS>
__global__ __launch_bounds__(256) void gn_bwd_apply_kernel(GNSrc s, const bf16_t* dy, const float* mean_rstd, const float* gsum,
                                                           const float* gamma, const float* beta, const bf16_t* add1,
                                                           const bf16_t* add2, bf16_t* dx1, bf16_t* dx2, int HW, int G,
                                                           int pix_per_chunk, int silu, float inv_n, const float* partial) {
  const int C = s.C1 + s.C2, cpg = C / G;
  const GNMap m(C);
  const int b = blockIdx.y;
  __shared__ float gs_lds[512];
  __shared__ float mr_lds[512];
  if (!gsum) gn_block_sum_partials(partial + (size_t)b * gridDim.x * G * 2, gridDim.x, G,
                                   [&](int g, double sm, double sq) { gs_lds[g * 2] = (float)sm; gs_lds[g * 2 + 1] = (float)sq; });
  else {
    for (int i = threadIdx.x; i < G * 2; i += 256) gs_lds[i] = gsum[(size_t)b * G * 2 + i];
  }
  gn_stage_stats(mr_lds, mean_rstd + (size_t)b * G * 2, G);
  const int p0 = blockIdx.x * pix_per_chunk;
  int p1 = p0 + pix_per_chunk; if (p1 > HW) p1 = HW;
  if (p0 >= p1) return;
  if (silu) gn_bwd_apply_body<NS, true>(s, m, b, HW, C, cpg, p0, p1, mr_lds, gs_lds, inv_n, dy, gamma, beta, add1, add2, dx1, dx2, silu);
  else gn_bwd_apply_body<NS, false>(s, m, b, HW, C, cpg, p0, p1, mr_lds, gs_lds, inv_n, dy, gamma, beta, add1, add2, dx1, dx2, silu);
}

// ------------------------------------------------------------------------------------------------
// Small maps: one workgroup per (batch, group) SLAB, everything in one launch.
// At the 8x8 / 16x16 levels a GroupNorm moves 5-20 MB; the chunked kernels above need two launches each way (statistics, then
// apply with the partials finalised in its prologue) and each is launch latency plus two dependent memory round trips:
// 8 + 15 us forward and 10 + 12 us backward for 5 MB at 8x8 x 1280 (profiles/r02_roofline_per_shape.csv), ~70 such pairs per
// step.  A slab of HW x (C/G) <= 10240 elements (<= 40 per thread) fits in registers: load once, reduce over the block in double
// (same finalisation as gn_finalize_kernel), normalise, store.  Requirements (else the chunked path): C/G % 4 == 0 (8-byte
// quads), no group straddling the two concat sources, no per-channel parameter gradients wanted in the backward.
// NI = quads per thread (template: the loads of all NI quads are issued before the first use).
// ------------------------------------------------------------------------------------------------
constexpr int GN_SLAB_MAX = 10240;

struct GNSlab {
  const bf16_t* x;      // source holding this group: x + pix * ldx + cx
  int ldx, cx;          // row stride of that source, first channel of the group inside it
  int c0;               // first channel of the group in the concatenated numbering (gamma / beta / y / dy index)
  bool first;
  __device__ GNSlab(const GNSrc& s, int g, int cpg) {
    c0 = g * cpg;
    first = c0 < s.C1;
    x = first ? s.x1 : s.x2;
    ldx = first ? s.C1 : s.C2;
    cx = first ? c0 : c0 - s.C1;
  }
};
__device__ __forceinline__ double gn_slab_block_sum(double v, double* red) {      // all 256 threads; result broadcast
  v = wave_sum_f64(v);
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = v;
  __syncthreads();
  const double r = (red[0] + red[1]) + (red[2] + red[3]);
  __syncthreads();
  return r;
}

template <int NI, bool SILU>
__global__ __launch_bounds__(256) void gn_slab_fwd_kernel(GNSrc s, const float* gamma, const float* beta, bf16_t* y, float* mean_rstd,
                                                          int HW, int G, float inv_n, float eps) {
  __shared__ double red[4];
  const int C = s.C1 + s.C2, cpg = C / G, q4 = cpg >> 2, items = HW * q4;
  const int b = blockIdx.y, g = blockIdx.x;
  const GNSlab sl(s, g, cpg);
  int row[NI], qd[NI];
  bool in[NI];
  uint2 raw[NI];
#pragma unroll
  for (int k = 0; k < NI; ++k) {
    const int it = threadIdx.x + k * 256;
    in[k] = it < items;
    const int ic = in[k] ? it : items - 1;
    row[k] = ic / q4; qd[k] = (ic - row[k] * q4) * 4;
    raw[k] = *(const uint2*)(sl.x + ((size_t)b * HW + row[k]) * sl.ldx + sl.cx + qd[k]);
  }
  float ga[NI][4], be[NI][4];
#pragma unroll
  for (int k = 0; k < NI; ++k) {
    const float4 g4 = *(const float4*)(gamma + sl.c0 + qd[k]), b4 = *(const float4*)(beta + sl.c0 + qd[k]);
    ga[k][0] = g4.x; ga[k][1] = g4.y; ga[k][2] = g4.z; ga[k][3] = g4.w;
    be[k][0] = b4.x; be[k][1] = b4.y; be[k][2] = b4.z; be[k][3] = b4.w;
  }
  float sm = 0.f, sq = 0.f;
#pragma unroll
  for (int k = 0; k < NI; ++k) {
    float f[4];
    unpack4(raw[k], f);
    const float w = in[k] ? 1.f : 0.f;
#pragma unroll
    for (int j = 0; j < 4; ++j) { const float v = f[j] * w; sm += v; sq += v * v; }
  }
  const double S = gn_slab_block_sum((double)sm, red), Q = gn_slab_block_sum((double)sq, red);
  const double mean_d = S * inv_n;
  double var = Q * inv_n - mean_d * mean_d;
  if (var < 0.0) var = 0.0;
  const float mean = (float)mean_d, rstd = (float)(1.0 / sqrt(var + (double)eps));
  if (threadIdx.x == 0) { mean_rstd[((size_t)b * G + g) * 2] = mean; mean_rstd[((size_t)b * G + g) * 2 + 1] = rstd; }
#pragma unroll
  for (int k = 0; k < NI; ++k) {
    float f[4];
    unpack4(raw[k], f);
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const float sc = rstd * ga[k][j];
      const float z = f[j] * sc + (be[k][j] - mean * sc);
      f[j] = SILU ? silu_f(z) : z;
    }
    if (in[k]) *(uint2*)(y + ((size_t)b * HW + row[k]) * C + sl.c0 + qd[k]) = pack4(f);
  }
}

// dx = rstd * (dz*gamma - (S1 + xhat*S2)/n) (+ add) with S1 = sum gamma*dz, S2 = sum gamma*dz*xhat over the slab
template <int NI, bool SILU>
__global__ __launch_bounds__(256) void gn_slab_bwd_kernel(GNSrc s, const bf16_t* dy, const float* mean_rstd, const float* gamma,
                                                          const float* beta, const bf16_t* add1, const bf16_t* add2, bf16_t* dx1,
                                                          bf16_t* dx2, int HW, int G, float inv_n) {
  __shared__ double red[4];
  const int C = s.C1 + s.C2, cpg = C / G, q4 = cpg >> 2, items = HW * q4;
  const int b = blockIdx.y, g = blockIdx.x;
  const GNSlab sl(s, g, cpg);
  const bf16_t* add = sl.first ? add1 : add2;
  bf16_t* dx = sl.first ? dx1 : dx2;
  const float wa = add ? 1.f : 0.f;
  const bf16_t* addp = add ? add : sl.x;          // no shortcut gradient: re-read x (same cache lines as rx), weighted by wa = 0
  // (per-item offsets are kept as two 32-bit element offsets: the {row, quad, in-range} arrays of the first version were left in
  // scratch memory by the compiler — 32 / 48 / 96 bytes per lane, profiles/r03_isa_resources.txt — although every index is static)
  unsigned ox[NI], oc[NI];          // element offset of the quad in x / add / dx (row stride ldx) and its channel offset in the group
  uint2 rx[NI], rd[NI], ra[NI];
#pragma unroll
  for (int k = 0; k < NI; ++k) {
    const int it = threadIdx.x + k * 256;
    const int ic = it < items ? it : items - 1;
    const int row = ic / q4, qd = (ic - row * q4) * 4;
    const size_t pix = (size_t)b * HW + row;
    ox[k] = (unsigned)(pix * sl.ldx + sl.cx + qd);
    oc[k] = (unsigned)qd;
    rx[k] = *(const uint2*)(sl.x + ox[k]);
    rd[k] = *(const uint2*)(dy + pix * C + sl.c0 + qd);
    ra[k] = *(const uint2*)(addp + ox[k]);                   // unconditional (a conditional load left ra[] in scratch memory: 8 NI bytes per lane)
  }
  const float2 mr = *(const float2*)(mean_rstd + ((size_t)b * G + g) * 2);
  const float mean = mr.x, rstd = mr.y;
  float xh[NI][4], gd[NI][4];      // xhat and gamma * dz
  float s1 = 0.f, s2 = 0.f;
#pragma unroll
  for (int k = 0; k < NI; ++k) {
    const float4 g4 = *(const float4*)(gamma + sl.c0 + oc[k]), b4 = *(const float4*)(beta + sl.c0 + oc[k]);
    const float ga[4] = {g4.x, g4.y, g4.z, g4.w}, be[4] = {b4.x, b4.y, b4.z, b4.w};
    float f[4], d[4];
    unpack4(rx[k], f);
    unpack4(rd[k], d);
    const float w = (int)(threadIdx.x + k * 256) < items ? 1.f : 0.f;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      xh[k][j] = (f[j] - mean) * rstd;
      float dz = d[j];
      if (SILU) dz *= dsilu_f(ga[j] * xh[k][j] + be[j]);
      gd[k][j] = dz * ga[j];
      s1 += gd[k][j] * w; s2 += gd[k][j] * xh[k][j] * w;
    }
  }
  const float g1 = (float)gn_slab_block_sum((double)s1, red) * inv_n, g2 = (float)gn_slab_block_sum((double)s2, red) * inv_n;
#pragma unroll
  for (int k = 0; k < NI; ++k) {
    float a[4], o[4];
    unpack4(ra[k], a);
#pragma unroll
    for (int j = 0; j < 4; ++j) o[j] = rstd * (gd[k][j] - g1 - xh[k][j] * g2) + a[j] * wa;
    if ((int)(threadIdx.x + k * 256) < items) *(uint2*)(dx + ox[k]) = pack4(o);
  }
}

// quads per thread of the slab kernels for this shape, 0 = not eligible
int gn_slab_ni(int C1, int C2, int HW, int G) {
  const bool off = false;
  const int C = C1 + C2, cpg = C / G;
  if (off || cpg % 4 != 0 || (C2 > 0 && C1 % cpg != 0) || (long long)HW * cpg > GN_SLAB_MAX) return 0;
  const int ni = cdiv(HW * (cpg / 4), 256);
  return ni <= 3 ? 3 : ni <= 5 ? 5 : 10;
}


template <int NI>
void gn_slab_fwd_launch(const GNSrc& s, const float* gamma, const float* beta, bf16_t* y, float* mean_rstd, int Bn, int HW, int G, float eps,
                        int silu, hipStream_t st) {
  const float inv_n = 1.f / ((float)((s.C1 + s.C2) / G) * (float)HW);
  if (silu) hipLaunchKernelGGL((gn_slab_fwd_kernel<NI, true>), dim3(G, Bn), dim3(256), 0, st, s, gamma, beta, y, mean_rstd, HW, G, inv_n, eps);
  else hipLaunchKernelGGL((gn_slab_fwd_kernel<NI, false>), dim3(G, Bn), dim3(256), 0, st, s, gamma, beta, y, mean_rstd, HW, G, inv_n, eps);
}
int gn_slab_fwd(int ni, const GNSrc& s, const float* gamma, const float* beta, bf16_t* y, float* mean_rstd, int Bn, int HW, int G, float eps,
                int silu, hipStream_t st) {
  E4T_LOG_LAUNCH("gn_slab_fwd_kernel<%d, %s>|B%d HW%d C%d G%d|%.0f|0", ni, silu ? "true" : "false", Bn, HW, s.C1 + s.C2, G,
                 4.0 * Bn * (double)HW * (s.C1 + s.C2));
  if (ni == 3) gn_slab_fwd_launch<3>(s, gamma, beta, y, mean_rstd, Bn, HW, G, eps, silu, st);
  else if (ni == 5) gn_slab_fwd_launch<5>(s, gamma, beta, y, mean_rstd, Bn, HW, G, eps, silu, st);
  else gn_slab_fwd_launch<10>(s, gamma, beta, y, mean_rstd, Bn, HW, G, eps, silu, st);
  E4T_CHECK_LAUNCH("gn_slab_fwd_kernel");
  return 0;
}
template <int NI>
void gn_slab_bwd_launch(const GNSrc& s, const bf16_t* dy, const float* mean_rstd, const float* gamma, const float* beta, const bf16_t* add1,
                        const bf16_t* add2, bf16_t* dx1, bf16_t* dx2, int Bn, int HW, int G, int silu, hipStream_t st) {
  const float inv_n = 1.f / ((float)((s.C1 + s.C2) / G) * (float)HW);
  if (silu) hipLaunchKernelGGL((gn_slab_bwd_kernel<NI, true>), dim3(G, Bn), dim3(256), 0, st, s, dy, mean_rstd, gamma, beta, add1, add2, dx1, dx2, HW, G, inv_n);
  else hipLaunchKernelGGL((gn_slab_bwd_kernel<NI, false>), dim3(G, Bn), dim3(256), 0, st, s, dy, mean_rstd, gamma, beta, add1, add2, dx1, dx2, HW, G, inv_n);
}

size_t gn_lds_bytes(int C) {
  const int QW = C / 8;
  const int PG = QW <= 256 ? 256 / QW : 1;
  return ((size_t)PG * C * 2 + (size_t)C * 2) * sizeof(float);
}

// ------------------------------------------------------------------------------------------------
// LayerNorm over the last dim D (D % 8 == 0, D <= 3*64*8 = 1536): one wave per row.
// ------------------------------------------------------------------------------------------------
constexpr int LN_MAXC = 3;

// Every load of a row (x, gamma, beta / x, dy, gamma, add) is issued up front from one basic block: chunk indices past the row end are
// clamped to the last chunk (loads unconditional, contributions and stores masked) — the guarded loads of the first version each
// cost a full memory round trip (3 + 3 in a row for D = 1280: 17 us for the ViT's 4112-row LayerNorms at 1.2 TB/s).
// One wave normalises R rows at once (a block = 4 waves = 4R rows): the forward has a single 2-byte stream to read, so with one
// row per wave the bytes in flight per CU (occupancy x one 16-B load per lane) cover only about a third of the HBM
// latency-bandwidth product — measured 3.0 TB/s algorithmic at M65536 D320 against 5.9 TB/s for the backward of the same
// tensor, which has three streams in flight.  R rows = R independent load / reduce chains per wave.
// XF32: the input rows are fp32 (the CLIP-ViT's fp32 residual stream); the output stays bf16 (the next GEMM's operand).
template <bool XF32>
struct LnChunk {                  // 8 consecutive elements of a row, as loaded
  uint4 a, b;
  __device__ __forceinline__ void load(const void* x, size_t elem) {
    if (XF32) { const uint4* q = (const uint4*)((const float*)x + elem); a = q[0]; b = q[1]; }
    else a = *(const uint4*)((const bf16_t*)x + elem);
  }
  __device__ __forceinline__ void get(float* v) const {
    if (XF32) {
      v[0] = __uint_as_float(a.x); v[1] = __uint_as_float(a.y); v[2] = __uint_as_float(a.z); v[3] = __uint_as_float(a.w);
      v[4] = __uint_as_float(b.x); v[5] = __uint_as_float(b.y); v[6] = __uint_as_float(b.z); v[7] = __uint_as_float(b.w);
    } else unpack8(a, v);
  }
};
template <int NC, int R, bool XF32 = false>
__global__ __launch_bounds__(256) void ln_fwd_kernel(const void* x, const float* gamma, const float* beta, bf16_t* y,
                                                     float* mean_rstd, int M, int D, float eps) {
  const int row0 = (blockIdx.x * 4 + (threadIdx.x >> 6)) * R, lane = threadIdx.x & 63;
  if (row0 >= M) return;
  const int nc = D >> 3;
  LnChunk<XF32> raw[R][NC];
  float ga[NC][8], be[NC][8];
  int cc[NC], rr[R];
  bool act[NC];
#pragma unroll
  for (int i = 0; i < NC; ++i) {
    const int c = lane + i * 64;
    act[i] = c < nc;
    cc[i] = act[i] ? c : nc - 1;
  }
#pragma unroll
  for (int r = 0; r < R; ++r) {
    rr[r] = row0 + r < M ? row0 + r : M - 1;          // ragged last wave: re-read the last row, store nothing
#pragma unroll
    for (int i = 0; i < NC; ++i) raw[r][i].load(x, (size_t)rr[r] * D + cc[i] * 8);
  }
#pragma unroll
  for (int i = 0; i < NC; ++i) { ld8f(gamma + cc[i] * 8, ga[i]); ld8f(beta + cc[i] * 8, be[i]); }
  float mean[R], rstd[R];
#pragma unroll
  for (int r = 0; r < R; ++r) {
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < NC; ++i) {
      float v[8];
      raw[r][i].get(v);
      if (act[i])
#pragma unroll
        for (int j = 0; j < 8; ++j) s += v[j];
    }
    mean[r] = s;
  }
#pragma unroll
  for (int r = 0; r < R; ++r) mean[r] = wave_sum(mean[r]) / D;
#pragma unroll
  for (int r = 0; r < R; ++r) {
    float q = 0.f;
#pragma unroll
    for (int i = 0; i < NC; ++i) {
      float v[8];
      raw[r][i].get(v);
      if (act[i])
#pragma unroll
        for (int j = 0; j < 8; ++j) { const float d = v[j] - mean[r]; q += d * d; }
    }
    rstd[r] = q;
  }
#pragma unroll
  for (int r = 0; r < R; ++r) rstd[r] = rsqrtf(wave_sum(rstd[r]) / D + eps);
#pragma unroll
  for (int r = 0; r < R; ++r) {
    const bool live = row0 + r < M;
    if (lane == 0 && mean_rstd && live) { mean_rstd[(size_t)rr[r] * 2] = mean[r]; mean_rstd[(size_t)rr[r] * 2 + 1] = rstd[r]; }
#pragma unroll
    for (int i = 0; i < NC; ++i) {
      float v[8], o[8];
      raw[r][i].get(v);
#pragma unroll
      for (int j = 0; j < 8; ++j) o[j] = (v[j] - mean[r]) * rstd[r] * ga[i][j] + be[i][j];
      if (act[i] && live) *(uint4*)(y + (size_t)rr[r] * D + cc[i] * 8) = pack8(o);
    }
  }
}
template <int NC>
__global__ __launch_bounds__(256) void ln_bwd_kernel(const bf16_t* x, const bf16_t* dy, const float* gamma, const float* mean_rstd,
                                                     const bf16_t* add, bf16_t* dx, int M, int D) {
  const int row = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
  if (row >= M) return;
  const int nc = D >> 3;
  uint4 rx[NC], rd[NC], ra[NC];
  float ga[NC][8];
  int cc[NC];
  bool act[NC];
  const bf16_t* asrc = add ? add : dy;                       // no residual gradient: alias dy (ignored below), the load stays unconditional
  const float wa = add ? 1.f : 0.f;
#pragma unroll
  for (int i = 0; i < NC; ++i) {
    const int c = lane + i * 64;
    act[i] = c < nc;
    cc[i] = act[i] ? c : nc - 1;
    const size_t o = (size_t)row * D + cc[i] * 8;
    rx[i] = *(const uint4*)(x + o);
    rd[i] = *(const uint4*)(dy + o);
    ra[i] = *(const uint4*)(asrc + o);
  }
  const float2 mr = *(const float2*)(mean_rstd + (size_t)row * 2);
  const float mean = mr.x, rstd = mr.y;
#pragma unroll
  for (int i = 0; i < NC; ++i) ld8f(gamma + cc[i] * 8, ga[i]);
  float xh[NC][8], dg[NC][8];
  float s1 = 0.f, s2 = 0.f;
#pragma unroll
  for (int i = 0; i < NC; ++i) {
    float xv[8], dv[8];
    unpack8(rx[i], xv);
    unpack8(rd[i], dv);
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      xh[i][j] = (xv[j] - mean) * rstd;
      dg[i][j] = dv[j] * ga[i][j];
      if (act[i]) { s1 += dg[i][j]; s2 += dg[i][j] * xh[i][j]; }
    }
  }
  s1 = wave_sum(s1) / D; s2 = wave_sum(s2) / D;
#pragma unroll
  for (int i = 0; i < NC; ++i) {
    float o[8], a[8];
    unpack8(ra[i], a);
#pragma unroll
    for (int j = 0; j < 8; ++j) o[j] = rstd * (dg[i][j] - s1 - xh[i][j] * s2) + a[j] * wa;
    if (act[i]) *(uint4*)(dx + (size_t)row * D + cc[i] * 8) = pack8(o);
  }
}

// column sums for norm parameter gradients: out[c] (+)= sum_r f(r, c); one thread per column, rows split over grid.y
// mode 0: dbeta = sum dy ; mode 1: dgamma = sum dy * (x - mean_r) * rstd_r
// ------------------------------------------------------------------------------------------------
// Column reductions over the rows of a (M, C) bf16 matrix — bias gradients and LayerNorm parameter gradients:
//   MODE 0:  out0[c] = sum_r x[r][c]
//   MODE 1:  out0[c] = sum_r dy[r][c] * (x[r][c] - mean_r) * rstd_r   (dgamma),   out1[c] = sum_r dy[r][c]   (dbeta)
// Stage 1: grid (C/64 column blocks, row splits); 256 threads = 32 row lanes x 8 chunks of 8 columns, 16-byte loads,
// fixed-order LDS reduction over the row lanes -> partial[split][C].  Stage 2: one thread per column adds the splits in
// order (deterministic) and stores or accumulates.  (The first version walked 512 rows per thread with 2-byte loads:
// 165 us for 65536 x 320; this one is bandwidth-bound.)
// ------------------------------------------------------------------------------------------------
template <int MODE>
__global__ __launch_bounds__(256) void colreduce_kernel(const bf16_t* x, int ldx, const bf16_t* dy, const float* mean_rstd, int M, int C,
                                                        int rows_per_split, float* part0, float* part1) {
  __shared__ float red[2][32][65];
  const int rl = threadIdx.x >> 3, ck = threadIdx.x & 7;
  const int c0 = blockIdx.x * 64 + ck * 8;
  const int r0 = blockIdx.y * rows_per_split;
  int r1 = r0 + rows_per_split; if (r1 > M) r1 = M;
  float a[8], b[8];
#pragma unroll
  for (int j = 0; j < 8; ++j) a[j] = b[j] = 0.f;
  if (c0 < C) {
    for (int r = r0 + rl; r < r1; r += 32) {
      float xv[8];
      unpack8(*(const uint4*)(x + (size_t)r * ldx + c0), xv);
      if (MODE == 0) {
#pragma unroll
        for (int j = 0; j < 8; ++j) a[j] += xv[j];
      } else {
        float dv[8];
        unpack8(*(const uint4*)(dy + (size_t)r * ldx + c0), dv);
        const float mean = mean_rstd[(size_t)r * 2], rstd = mean_rstd[(size_t)r * 2 + 1];
#pragma unroll
        for (int j = 0; j < 8; ++j) { a[j] += dv[j] * ((xv[j] - mean) * rstd); b[j] += dv[j]; }
      }
    }
  }
#pragma unroll
  for (int j = 0; j < 8; ++j) { red[0][rl][ck * 8 + j] = a[j]; if (MODE == 1) red[1][rl][ck * 8 + j] = b[j]; }
  __syncthreads();
  const int q = threadIdx.x >> 6, col = threadIdx.x & 63;      // q = 0: first quantity, q = 1: second (MODE 1 only)
  if (q <= MODE && blockIdx.x * 64 + col < C) {
    float t = 0.f;
#pragma unroll 8
    for (int i = 0; i < 32; ++i) t += red[q][i][col];
    (q == 0 ? part0 : part1)[(size_t)blockIdx.y * C + blockIdx.x * 64 + col] = t;
  }
}
__global__ __launch_bounds__(256) void colreduce_final_kernel(const float* part, int nsplit, int C, float* out, int accumulate) {
  const int c = blockIdx.x * 256 + threadIdx.x;
  if (c >= C) return;
  float t = 0.f;
  for (int i = 0; i < nsplit; ++i) t += part[(size_t)i * C + c];
  out[c] = accumulate ? out[c] + t : t;
}
int colreduce_splits(int M) { int n = cdiv(M, 1024); return n > 128 ? 128 : (n < 1 ? 1 : n); }

int gn_chunks(int Bn, int HW) {
  const int target = 1024;      // workgroups per launch (swept in round 4: profiles/r04_ab/r04f_*)
  int ch = target / (Bn > 0 ? Bn : 1);
  if (ch < 1) ch = 1;
  const int maxch = HW / 8 > 0 ? HW / 8 : 1;
  if (ch > maxch) ch = maxch;
  if (ch > 256) ch = 256;
  return ch;
}

}  // namespace

// one 16-byte chunk per thread up to C = 2048, two above (GNMap): the kernels are instantiated per slot count
#define GN_TWO_SLOTS ((C1 + C2) / 8 > 256)

extern "C" int e4t_groupnorm_num_chunks(int Bn, int HW) { return gn_chunks(Bn, HW); }

extern "C" size_t e4t_groupnorm_workspace_bytes(int Bn, int HW, int C, int G, int with_param_grads) {
  const size_t ch = (size_t)gn_chunks(Bn, HW);
  size_t b = (size_t)Bn * ch * G * 2 * sizeof(float);
  if (with_param_grads) b += (size_t)Bn * ch * C * 2 * sizeof(float);
  return b;
}

static int gn_check(const void* x1, int C1, const void* x2, int C2, int Bn, int HW, int G) {
  const int C = C1 + C2;
  E4T_REQUIRE(x1 && C1 > 0 && Bn > 0 && HW > 0 && G > 0, "groupnorm: bad arguments");
  E4T_REQUIRE((x2 != nullptr) == (C2 > 0), "groupnorm: x2/C2 mismatch");
  E4T_REQUIRE(C % G == 0 && C1 % 8 == 0 && C2 % 8 == 0, "groupnorm: C=%d must divide into G=%d groups, C1/C2 %% 8 == 0", C, G);
  E4T_REQUIRE(C <= GN_MAXS * 256 * 8 && G <= 256, "groupnorm: C=%d too large", C);
  return 0;
}

extern "C" int e4t_groupnorm_stats(const void* x1, int C1, const void* x2, int C2, int Bn, int HW, int G, float eps,
                                   float* mean_rstd, void* workspace, size_t ws_bytes, e4t_stream stream) {
  if (int e = gn_check(x1, C1, x2, C2, Bn, HW, G)) return e;
  const int C = C1 + C2, ch = gn_chunks(Bn, HW), ppc = cdiv(HW, ch);
  E4T_REQUIRE(workspace && ws_bytes >= (size_t)Bn * ch * G * 2 * sizeof(float), "groupnorm_stats: workspace too small");
  GNSrc s{(const bf16_t*)x1, (const bf16_t*)x2, C1, C2};
  hipStream_t st = (hipStream_t)stream;
  E4T_LOG_LAUNCH("gn_stats_kernel<%d>|B%d HW%d C%d G%d|%.0f|0", GN_TWO_SLOTS ? 2 : 1, Bn, HW, C, G, 2.0 * Bn * (double)HW * C);
  if (GN_TWO_SLOTS) hipLaunchKernelGGL((gn_stats_kernel<2>), dim3(ch, Bn), dim3(256), gn_lds_bytes(C), st, s, HW, G, ppc, (float*)workspace);
  else hipLaunchKernelGGL((gn_stats_kernel<1>), dim3(ch, Bn), dim3(256), gn_lds_bytes(C), st, s, HW, G, ppc, (float*)workspace);
  E4T_CHECK_LAUNCH("gn_stats_kernel");
  const int BG = Bn * G;
  hipLaunchKernelGGL(gn_finalize_kernel, dim3(cdiv(BG, 4)), dim3(256), 0, st, (const float*)workspace, ch, G, BG,
                     1.f / ((float)(C / G) * (float)HW), eps, mean_rstd);
  E4T_CHECK_LAUNCH("gn_finalize_kernel");
  return 0;
}

extern "C" int e4t_groupnorm_apply(const void* x1, int C1, const void* x2, int C2, const float* mean_rstd, const float* gamma,
                                   const float* beta, void* y, int Bn, int HW, int G, int silu, e4t_stream stream) {
  if (int e = gn_check(x1, C1, x2, C2, Bn, HW, G)) return e;
  E4T_REQUIRE(mean_rstd && gamma && beta && y, "groupnorm_apply: null argument");
  const int ch = gn_chunks(Bn, HW), ppc = cdiv(HW, ch);
  GNSrc s{(const bf16_t*)x1, (const bf16_t*)x2, C1, C2};
  E4T_LOG_LAUNCH("gn_apply_kernel<false, %d>|B%d HW%d C%d G%d|%.0f|0", GN_TWO_SLOTS ? 2 : 1, Bn, HW, C1 + C2, G, 4.0 * Bn * (double)HW * (C1 + C2));
  if (GN_TWO_SLOTS) hipLaunchKernelGGL((gn_apply_kernel<false, 2>), dim3(ch, Bn), dim3(256), 0, (hipStream_t)stream, s, mean_rstd, gamma, beta, (bf16_t*)y, HW, G, ppc,
                     silu, (const float*)nullptr, 0, 0.f, 0.f, (float*)nullptr);
  else hipLaunchKernelGGL((gn_apply_kernel<false, 1>), dim3(ch, Bn), dim3(256), 0, (hipStream_t)stream, s, mean_rstd, gamma, beta, (bf16_t*)y, HW, G, ppc,
                     silu, (const float*)nullptr, 0, 0.f, 0.f, (float*)nullptr);
  E4T_CHECK_LAUNCH("gn_apply_kernel");
  return 0;
}

// Chunk partials (the format gn_stats_kernel writes: partial[b][chunk][g][2]) from the column statistics the producing GEMM /
// conv epilogue left behind (gemm.hip, write_tile): cs[blk][c][2] = (sum, sum of squares) of channel c over the 32 pixels of
// block blk.  One workgroup per (chunk of 32-pixel blocks, batch entry): thread t owns columns t, t+256, ...; a few KB .. MB of
// statistics are read instead of a pass over the whole activation.  Deterministic (fixed summation order).
namespace {
__global__ __launch_bounds__(256) void gn_stats_cols_kernel(const float* cs1, int C1, const float* cs2, int C2, int nblk_img, int blk_per_chunk,
                                                            int G, float* partial) {
  extern __shared__ float lds[];            // [C][2] column totals of this chunk
  const int C = C1 + C2, cpg = C / G;
  const int b = blockIdx.y, chunk = blockIdx.x;
  const size_t rb0 = (size_t)b * nblk_img + (size_t)chunk * blk_per_chunk;
  for (int c = threadIdx.x; c < C; c += 256) {
    const float* src = c < C1 ? cs1 + (rb0 * C1 + c) * 2 : cs2 + (rb0 * C2 + (c - C1)) * 2;
    const size_t stride = (size_t)(c < C1 ? C1 : C2) * 2;
    float sm = 0.f, sq = 0.f;
    for (int k = 0; k < blk_per_chunk; ++k) {
      const float2 v = *(const float2*)(src + k * stride);
      sm += v.x; sq += v.y;
    }
    lds[c * 2] = sm; lds[c * 2 + 1] = sq;
  }
  __syncthreads();
  if ((int)threadIdx.x < G) {
    float ta = 0.f, tb = 0.f;
    for (int c = threadIdx.x * cpg; c < (threadIdx.x + 1) * cpg; ++c) { ta += lds[c * 2]; tb += lds[c * 2 + 1]; }
    float* o = partial + (((size_t)b * gridDim.x + chunk) * G + threadIdx.x) * 2;
    o[0] = ta; o[1] = tb;
  }
}
}  // namespace

extern "C" int e4t_groupnorm_fwd_cs(const void* x1, int C1, const float* cs1, const void* x2, int C2, const float* cs2, const float* gamma,
                                    const float* beta, void* y, float* mean_rstd, int Bn, int HW, int G, float eps, int silu,
                                    void* workspace, size_t ws_bytes, e4t_stream stream) {
  if (int e = gn_check(x1, C1, x2, C2, Bn, HW, G)) return e;
  E4T_REQUIRE(cs1 && ((cs2 != nullptr) == (C2 > 0)) && gamma && beta && y && mean_rstd && HW % 32 == 0, "groupnorm_fwd_cs: bad arguments");
  if (const int ni = gn_slab_ni(C1, C2, HW, G)) {      // small map: one launch, the column statistics are not needed
    GNSrc ss{(const bf16_t*)x1, (const bf16_t*)x2, C1, C2};
    return gn_slab_fwd(ni, ss, gamma, beta, (bf16_t*)y, mean_rstd, Bn, HW, G, eps, silu, (hipStream_t)stream);
  }
  const int C = C1 + C2, ch = gn_chunks(Bn, HW), ppc = cdiv(HW, ch), nblk = HW / 32;
  const int npart = ch < nblk ? ch : nblk;                   // chunks of whole 32-pixel blocks
  E4T_REQUIRE(nblk % npart == 0, "groupnorm_fwd_cs: %d blocks do not split into %d chunks", nblk, npart);
  E4T_REQUIRE(workspace && ws_bytes >= (size_t)Bn * npart * G * 2 * sizeof(float), "groupnorm_fwd_cs: workspace too small");
  GNSrc s{(const bf16_t*)x1, (const bf16_t*)x2, C1, C2};
  hipStream_t st = (hipStream_t)stream;
  E4T_LOG_LAUNCH("gn_stats_cols_kernel|B%d HW%d C%d G%d|%.0f|0", Bn, HW, C, G, 8.0 * Bn * (double)nblk * C);
  hipLaunchKernelGGL(gn_stats_cols_kernel, dim3(npart, Bn), dim3(256), (size_t)C * 2 * sizeof(float), st, cs1, C1, cs2, C2, nblk, nblk / npart, G,
                     (float*)workspace);
  E4T_CHECK_LAUNCH("gn_stats_cols_kernel");
  E4T_LOG_LAUNCH("gn_apply_kernel<true, %d>|B%d HW%d C%d G%d silu%d|%.0f|0", GN_TWO_SLOTS ? 2 : 1, Bn, HW, C, G, silu, 4.0 * Bn * (double)HW * C);
  if (GN_TWO_SLOTS) hipLaunchKernelGGL((gn_apply_kernel<true, 2>), dim3(ch, Bn), dim3(256), 0, st, s, (const float*)nullptr, gamma, beta, (bf16_t*)y, HW, G, ppc, silu,
                     (const float*)workspace, npart, 1.f / ((float)(C / G) * (float)HW), eps, mean_rstd);
  else hipLaunchKernelGGL((gn_apply_kernel<true, 1>), dim3(ch, Bn), dim3(256), 0, st, s, (const float*)nullptr, gamma, beta, (bf16_t*)y, HW, G, ppc, silu,
                     (const float*)workspace, npart, 1.f / ((float)(C / G) * (float)HW), eps, mean_rstd);
  E4T_CHECK_LAUNCH("gn_apply_kernel");
  return 0;
}

extern "C" int e4t_groupnorm_fwd(const void* x1, int C1, const void* x2, int C2, const float* gamma, const float* beta, void* y,
                                 float* mean_rstd, int Bn, int HW, int G, float eps, int silu, void* workspace, size_t ws_bytes,
                                 e4t_stream stream) {
  if (int e = gn_check(x1, C1, x2, C2, Bn, HW, G)) return e;
  E4T_REQUIRE(gamma && beta && y && mean_rstd, "groupnorm_fwd: null argument");
  if (const int ni = gn_slab_ni(C1, C2, HW, G)) {
    GNSrc ss{(const bf16_t*)x1, (const bf16_t*)x2, C1, C2};
    return gn_slab_fwd(ni, ss, gamma, beta, (bf16_t*)y, mean_rstd, Bn, HW, G, eps, silu, (hipStream_t)stream);
  }
  const int C = C1 + C2, ch = gn_chunks(Bn, HW), ppc = cdiv(HW, ch);
  E4T_REQUIRE(workspace && ws_bytes >= (size_t)Bn * ch * G * 2 * sizeof(float), "groupnorm_fwd: workspace too small");
  GNSrc s{(const bf16_t*)x1, (const bf16_t*)x2, C1, C2};
  hipStream_t st = (hipStream_t)stream;
  E4T_LOG_LAUNCH("gn_stats_kernel<%d>|B%d HW%d C%d G%d|%.0f|0", GN_TWO_SLOTS ? 2 : 1, Bn, HW, C, G, 2.0 * Bn * (double)HW * C);
  if (GN_TWO_SLOTS) hipLaunchKernelGGL((gn_stats_kernel<2>), dim3(ch, Bn), dim3(256), gn_lds_bytes(C), st, s, HW, G, ppc, (float*)workspace);
  else hipLaunchKernelGGL((gn_stats_kernel<1>), dim3(ch, Bn), dim3(256), gn_lds_bytes(C), st, s, HW, G, ppc, (float*)workspace);
  E4T_CHECK_LAUNCH("gn_stats_kernel");
  E4T_LOG_LAUNCH("gn_apply_kernel<true, %d>|B%d HW%d C%d G%d silu%d|%.0f|0", GN_TWO_SLOTS ? 2 : 1, Bn, HW, C, G, silu, 4.0 * Bn * (double)HW * C);
  if (GN_TWO_SLOTS) hipLaunchKernelGGL((gn_apply_kernel<true, 2>), dim3(ch, Bn), dim3(256), 0, st, s, (const float*)nullptr, gamma, beta, (bf16_t*)y, HW, G, ppc, silu,
                     (const float*)workspace, ch, 1.f / ((float)(C / G) * (float)HW), eps, mean_rstd);
  else hipLaunchKernelGGL((gn_apply_kernel<true, 1>), dim3(ch, Bn), dim3(256), 0, st, s, (const float*)nullptr, gamma, beta, (bf16_t*)y, HW, G, ppc, silu,
                     (const float*)workspace, ch, 1.f / ((float)(C / G) * (float)HW), eps, mean_rstd);
  E4T_CHECK_LAUNCH("gn_apply_kernel");
  return 0;
}

extern "C" int e4t_groupnorm_bwd(const void* x1, int C1, const void* x2, int C2, const void* dy, const float* mean_rstd,
                                 const float* gamma, const float* beta, const void* add1, const void* add2, void* dx1, void* dx2,
                                 float* dgamma_dbeta_partial, int Bn, int HW, int G, int silu, void* workspace,
                                 size_t ws_bytes, e4t_stream stream) {
  if (int e = gn_check(x1, C1, x2, C2, Bn, HW, G)) return e;
  E4T_REQUIRE(dy && mean_rstd && gamma && beta && dx1 && ((dx2 != nullptr) == (C2 > 0)) && (!add2 || C2 > 0), "groupnorm_bwd: null argument");
  if (const int ni = dgamma_dbeta_partial ? 0 : gn_slab_ni(C1, C2, HW, G)) {      // (per-channel parameter gradients: chunked path)
    GNSrc ss{(const bf16_t*)x1, (const bf16_t*)x2, C1, C2};
    E4T_LOG_LAUNCH("gn_slab_bwd_kernel<%d, %s>|B%d HW%d C%d G%d add%d|%.0f|0", ni, silu ? "true" : "false", Bn, HW, C1 + C2, G,
                   (add1 != nullptr) + (add2 != nullptr), 2.0 * Bn * (double)HW * (3.0 * (C1 + C2) + (add1 ? C1 : 0) + (add2 ? C2 : 0)));
    if (ni == 3) gn_slab_bwd_launch<3>(ss, (const bf16_t*)dy, mean_rstd, gamma, beta, (const bf16_t*)add1, (const bf16_t*)add2, (bf16_t*)dx1, (bf16_t*)dx2, Bn, HW, G, silu, (hipStream_t)stream);
    else if (ni == 5) gn_slab_bwd_launch<5>(ss, (const bf16_t*)dy, mean_rstd, gamma, beta, (const bf16_t*)add1, (const bf16_t*)add2, (bf16_t*)dx1, (bf16_t*)dx2, Bn, HW, G, silu, (hipStream_t)stream);
    else gn_slab_bwd_launch<10>(ss, (const bf16_t*)dy, mean_rstd, gamma, beta, (const bf16_t*)add1, (const bf16_t*)add2, (bf16_t*)dx1, (bf16_t*)dx2, Bn, HW, G, silu, (hipStream_t)stream);
    E4T_CHECK_LAUNCH("gn_slab_bwd_kernel");
    return 0;
  }
  const int C = C1 + C2, ch = gn_chunks(Bn, HW), ppc = cdiv(HW, ch), BG = Bn * G;
  const size_t need = ((size_t)Bn * ch * G * 2 + (size_t)BG * 2) * sizeof(float);
  E4T_REQUIRE(workspace && ws_bytes >= need, "groupnorm_bwd: workspace too small (%zu < %zu)", ws_bytes, need);
  float* partial = (float*)workspace;
  float* gsum = partial + (size_t)Bn * ch * G * 2;
  GNSrc s{(const bf16_t*)x1, (const bf16_t*)x2, C1, C2};
  hipStream_t st = (hipStream_t)stream;
  E4T_LOG_LAUNCH("gn_bwd_stats_kernel<%d>|B%d HW%d C%d G%d|%.0f|0", GN_TWO_SLOTS ? 2 : 1, Bn, HW, C, G, 4.0 * Bn * (double)HW * C);
  E4T_LOG_LAUNCH("gn_bwd_apply_kernel<%d>|B%d HW%d C%d G%d add%d|%.0f|0", GN_TWO_SLOTS ? 2 : 1, Bn, HW, C, G, (add1 != nullptr) + (add2 != nullptr),
                 2.0 * Bn * (double)HW * (3.0 * C + (add1 ? C1 : 0) + (add2 ? C2 : 0)));
  if (GN_TWO_SLOTS) hipLaunchKernelGGL((gn_bwd_stats_kernel<2>), dim3(ch, Bn), dim3(256), gn_lds_bytes(C), st, s, (const bf16_t*)dy, mean_rstd,
                     gamma, beta, HW, G, ppc, silu, partial, dgamma_dbeta_partial);
  else hipLaunchKernelGGL((gn_bwd_stats_kernel<1>), dim3(ch, Bn), dim3(256), gn_lds_bytes(C), st, s, (const bf16_t*)dy, mean_rstd,
                     gamma, beta, HW, G, ppc, silu, partial, dgamma_dbeta_partial);
  E4T_CHECK_LAUNCH("gn_bwd_stats_kernel");
  (void)gsum;     // (S1, S2) are reduced from the partials inside gn_bwd_apply_kernel: no finalize launch
  if (GN_TWO_SLOTS) hipLaunchKernelGGL((gn_bwd_apply_kernel<2>), dim3(ch, Bn), dim3(256), 0, st, s, (const bf16_t*)dy, mean_rstd, (const float*)nullptr, gamma, beta,
                     (const bf16_t*)add1, (const bf16_t*)add2, (bf16_t*)dx1, (bf16_t*)dx2, HW, G, ppc, silu, 1.f / ((float)(C / G) * (float)HW),
                     (const float*)partial);
  else hipLaunchKernelGGL((gn_bwd_apply_kernel<1>), dim3(ch, Bn), dim3(256), 0, st, s, (const bf16_t*)dy, mean_rstd, (const float*)nullptr, gamma, beta,
                     (const bf16_t*)add1, (const bf16_t*)add2, (bf16_t*)dx1, (bf16_t*)dx2, HW, G, ppc, silu, 1.f / ((float)(C / G) * (float)HW),
                     (const float*)partial);
  E4T_CHECK_LAUNCH("gn_bwd_apply_kernel");
  return 0;
}

namespace {
int launch_ln_fwd(const void* x, bool xf32, const float* gamma, const float* beta, void* y, float* mean_rstd, int M, int D, float eps, e4t_stream stream) {
  E4T_REQUIRE(x && gamma && beta && y && M > 0, "layernorm_fwd: null argument");
  E4T_REQUIRE(D % 8 == 0 && D <= LN_MAXC * 64 * 8, "layernorm: D=%d must be a multiple of 8 and <= 1536", D);
  const int ncl = cdiv(D / 8, 64);      // 16-byte chunks per lane: the kernels are instantiated per count (no dead loads)
  // rows per wave: 4 / 2 / 2 for 1 / 2 / 3 chunks per lane; small M keeps one row per wave (enough blocks to fill the chip first)
  const int rpw = ((long long)M * ncl < 256 * 4 * 8 || xf32) ? 1 : ncl == 1 ? 4 : 2;
  E4T_LOG_LAUNCH("ln_fwd_kernel<%d, %d, %s>|M%d D%d|%.0f|0", ncl < 3 ? ncl : 3, rpw, xf32 ? "true" : "false", M, D, (xf32 ? 6.0 : 4.0) * (double)M * D);
#define E4T_LN_FWD(NC_, R_, F_) hipLaunchKernelGGL((ln_fwd_kernel<NC_, R_, F_>), dim3(cdiv(M, 4 * R_)), dim3(256), 0, (hipStream_t)stream, x, gamma, beta, (bf16_t*)y, mean_rstd, M, D, eps)
  if (xf32) { if (ncl == 1) E4T_LN_FWD(1, 1, true); else if (ncl == 2) E4T_LN_FWD(2, 1, true); else E4T_LN_FWD(3, 1, true); }
  else if (rpw == 1) { if (ncl == 1) E4T_LN_FWD(1, 1, false); else if (ncl == 2) E4T_LN_FWD(2, 1, false); else E4T_LN_FWD(3, 1, false); }
  else if (ncl == 1) E4T_LN_FWD(1, 4, false); else if (ncl == 2) E4T_LN_FWD(2, 2, false); else E4T_LN_FWD(3, 2, false);
#undef E4T_LN_FWD
  E4T_CHECK_LAUNCH("ln_fwd_kernel");
  return 0;
}
}  // namespace

extern "C" int e4t_layernorm_fwd(const void* x, const float* gamma, const float* beta, void* y, float* mean_rstd, int M, int D, float eps,
                                 e4t_stream stream) {
  return launch_ln_fwd(x, false, gamma, beta, y, mean_rstd, M, D, eps, stream);
}

extern "C" int e4t_layernorm_fwd_f32(const float* x, const float* gamma, const float* beta, void* y, float* mean_rstd, int M, int D, float eps,
                                     e4t_stream stream) {
  E4T_REQUIRE(((uintptr_t)x & 15) == 0, "layernorm_fwd_f32: x must be 16-byte aligned");
  return launch_ln_fwd(x, true, gamma, beta, y, mean_rstd, M, D, eps, stream);
}

extern "C" int e4t_layernorm_bwd(const void* x, const void* dy, const float* gamma, const float* mean_rstd, const void* add, void* dx,
                                 int M, int D, e4t_stream stream) {
  E4T_REQUIRE(x && dy && gamma && mean_rstd && dx && M > 0, "layernorm_bwd: null argument");
  E4T_REQUIRE(D % 8 == 0 && D <= LN_MAXC * 64 * 8, "layernorm: D=%d must be a multiple of 8 and <= 1536", D);
  const int ncl = cdiv(D / 8, 64);
  E4T_LOG_LAUNCH("ln_bwd_kernel<%d>|M%d D%d add%d|%.0f|0", ncl < 3 ? ncl : 3, M, D, add != nullptr, 2.0 * (double)M * D * (add ? 4 : 3));
#define E4T_LN_BWD(NC_) hipLaunchKernelGGL((ln_bwd_kernel<NC_>), dim3(cdiv(M, 4)), dim3(256), 0, (hipStream_t)stream, (const bf16_t*)x, (const bf16_t*)dy, gamma, mean_rstd, (const bf16_t*)add, (bf16_t*)dx, M, D)
  if (ncl == 1) E4T_LN_BWD(1); else if (ncl == 2) E4T_LN_BWD(2); else E4T_LN_BWD(3);
#undef E4T_LN_BWD
  E4T_CHECK_LAUNCH("ln_bwd_kernel");
  return 0;
}

// part_dgamma / part_dbeta: [nblk][D] fp32 with nblk = e4t_layernorm_param_grad_blocks(M); caller sums over dim 0.
extern "C" int e4t_colreduce_splits(int M) { return colreduce_splits(M); }

extern "C" int e4t_colsum(const void* x, int ldx, int M, int C, float* out, int accumulate, void* workspace, size_t ws_bytes,
                          e4t_stream stream) {
  E4T_REQUIRE(x && out && M > 0 && C > 0 && C % 8 == 0 && ldx % 8 == 0 && ((uintptr_t)x & 15) == 0, "colsum: bad arguments");
  const int ns = colreduce_splits(M);
  E4T_REQUIRE(workspace && ws_bytes >= (size_t)ns * C * sizeof(float), "colsum: workspace too small");
  hipStream_t st = (hipStream_t)stream;
  hipLaunchKernelGGL(colreduce_kernel<0>, dim3(cdiv(C, 64), ns), dim3(256), 0, st, (const bf16_t*)x, ldx, (const bf16_t*)nullptr,
                     (const float*)nullptr, M, C, cdiv(M, ns), (float*)workspace, (float*)nullptr);
  E4T_CHECK_LAUNCH("colreduce_kernel");
  hipLaunchKernelGGL(colreduce_final_kernel, dim3(cdiv(C, 256)), dim3(256), 0, st, (const float*)workspace, ns, C, out, accumulate);
  E4T_CHECK_LAUNCH("colreduce_final_kernel");
  return 0;
}

extern "C" int e4t_layernorm_param_grad(const void* x, const void* dy, const float* mean_rstd, int M, int D, float* dgamma, float* dbeta,
                                        int accumulate, void* workspace, size_t ws_bytes, e4t_stream stream) {
  E4T_REQUIRE(x && dy && mean_rstd && dgamma && dbeta && D % 8 == 0, "layernorm_param_grad: bad arguments");
  const int ns = colreduce_splits(M);
  E4T_REQUIRE(workspace && ws_bytes >= (size_t)2 * ns * D * sizeof(float), "layernorm_param_grad: workspace too small");
  float* p0 = (float*)workspace;
  float* p1 = p0 + (size_t)ns * D;
  hipStream_t st = (hipStream_t)stream;
  hipLaunchKernelGGL(colreduce_kernel<1>, dim3(cdiv(D, 64), ns), dim3(256), 0, st, (const bf16_t*)x, D, (const bf16_t*)dy, mean_rstd, M, D,
                     cdiv(M, ns), p0, p1);
  E4T_CHECK_LAUNCH("colreduce_kernel");
  hipLaunchKernelGGL(colreduce_final_kernel, dim3(cdiv(D, 256)), dim3(256), 0, st, (const float*)p0, ns, D, dgamma, accumulate);
  hipLaunchKernelGGL(colreduce_final_kernel, dim3(cdiv(D, 256)), dim3(256), 0, st, (const float*)p1, ns, D, dbeta, accumulate);
  E4T_CHECK_LAUNCH("colreduce_final_kernel");
  return 0;
}
