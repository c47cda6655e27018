// Weight-offset heads in closed form, grouped over a descriptor table (gfx950, HBM-bound).
//
// Reference: e4t/weightoffsets.py:5-23 evaluates, per attention projection and per UNet forward,
//   vx = linear1(v); vy = linear2(v); M = vx (x) vy; A = linear_column(M^T); Bm = linear_row(A^T); out = Bm^T
// i.e. two dense (dim x dim) GEMMs, and cross_attention.py:506,516,518 then forms W * (1 + out).
// Algebraically (verified against the reference class in tests/):
//   a = Wc.vx   b = Wr.vy   s = rowsum(Wr)
//   out[c][r] = a[r]*b[c] + bc[r]*s[c] + br[c]                       (c < col = d_out, r < row = d_in)
//   W_eff[c][r] = W[c][r] * (1 + out[c][r])
// which is O(row^2 + col^2 + row*col) memory traffic: Wc, Wr, W read once, W_eff written once
// (bf16, in both [d_out][d_in] and [d_in][d_out] layouts so forward and dX GEMMs are both "NT").
// Backward, with G = dW_eff o W:
//   dbr = rowsum_r(G)   db = G.a   ds = G.bc   dbc = G^T.s   da = G^T.b
//   dWc = da (x) vx   dvx = Wc^T.da   dWr = db (x) vy + ds (x) 1   dvy = Wr^T.db
//   dw1 = dvx*v  db1 = dvx  dw2 = dvy*v  db2 = dvy  dv = <dvx,w1> + <dvy,w2>   [dW = dW_eff o (1+out)]
//
// One launch covers every instance in the table (blockIdx.y = instance): the 96 WO heads of the
// SD UNet run forward in 2 launches and backward in 5, instead of 96 x (2 GEMMs + elementwise).
#include "common.h"
#include "../../include/e4t_hip.h"

namespace {

__device__ __forceinline__ float accw(float* p, float v, int accum) { return accum ? *p + v : v; }

// vecs layout (floats): a[row] b[col] s[col] vx[row] vy[col] da[row] db[col] ds[col]
__device__ __forceinline__ float* V_a(const e4t_wo_desc& d) { return d.vecs; }
__device__ __forceinline__ float* V_b(const e4t_wo_desc& d) { return d.vecs + d.row; }
__device__ __forceinline__ float* V_s(const e4t_wo_desc& d) { return d.vecs + d.row + d.col; }
__device__ __forceinline__ float* V_vx(const e4t_wo_desc& d) { return d.vecs + d.row + 2 * d.col; }
__device__ __forceinline__ float* V_vy(const e4t_wo_desc& d) { return d.vecs + 2 * d.row + 2 * d.col; }
__device__ __forceinline__ float* V_da(const e4t_wo_desc& d) { return d.vecs + 2 * d.row + 3 * d.col; }
__device__ __forceinline__ float* V_db(const e4t_wo_desc& d) { return d.vecs + 3 * d.row + 3 * d.col; }
__device__ __forceinline__ float* V_ds(const e4t_wo_desc& d) { return d.vecs + 3 * d.row + 4 * d.col; }

// ---- forward 1: a = Wc.vx, b = Wr.vy, s = rowsum(Wr); one wave per matrix row ----
__global__ __launch_bounds__(256) void wo_vec_kernel(const e4t_wo_desc* descs) {
  const e4t_wo_desc d = descs[blockIdx.y];
  const int i = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
  if (i >= d.row + d.col) return;
  const float v = d.v[0];
  if (i < d.row) {
    const float* wrow = d.wc + (size_t)i * d.row;
    float acc = 0.f;
    for (int k = lane * 4; k < d.row; k += 256) {
      const float4 w = *(const float4*)(wrow + k);
      const float4 w1 = *(const float4*)(d.w1 + k), b1 = *(const float4*)(d.b1 + k);
      const float x0 = w1.x * v + b1.x, x1 = w1.y * v + b1.y, x2 = w1.z * v + b1.z, x3 = w1.w * v + b1.w;
      acc += w.x * x0 + w.y * x1 + w.z * x2 + w.w * x3;
      if (i == 0) *(float4*)(V_vx(d) + k) = make_float4(x0, x1, x2, x3);
    }
    acc = wave_sum(acc);
    if (lane == 0) V_a(d)[i] = acc;
  } else {
    const int c = i - d.row;
    const float* wrow = d.wr + (size_t)c * d.col;
    float acc = 0.f, rs = 0.f;
    for (int k = lane * 4; k < d.col; k += 256) {
      const float4 w = *(const float4*)(wrow + k);
      const float4 w2 = *(const float4*)(d.w2 + k), b2 = *(const float4*)(d.b2 + k);
      const float y0 = w2.x * v + b2.x, y1 = w2.y * v + b2.y, y2 = w2.z * v + b2.z, y3 = w2.w * v + b2.w;
      acc += w.x * y0 + w.y * y1 + w.z * y2 + w.w * y3;
      rs += w.x + w.y + w.z + w.w;
      if (c == 0) *(float4*)(V_vy(d) + k) = make_float4(y0, y1, y2, y3);
    }
    acc = wave_sum(acc); rs = wave_sum(rs);
    if (lane == 0) { V_b(d)[c] = acc; V_s(d)[c] = rs; }
  }
}

// ---- forward 2: W_eff = W o (1 + out) -> bf16 [col][ld] and transposed bf16 [row][ldT]; 64x64 tiles ----
// Instances with wc == nullptr are plain weights: W_eff = W (cast + transposed copy only).
__global__ __launch_bounds__(256) void wo_apply_kernel(const e4t_wo_desc* descs) {
  const e4t_wo_desc d = descs[blockIdx.y];
  const int tr = (d.row + 63) >> 6, tc = (d.col + 63) >> 6;
  if ((int)blockIdx.x >= tr * tc) return;
  __shared__ bf16_t tile[64][66];
  const int c0 = (blockIdx.x / tr) * 64, r0 = (blockIdx.x % tr) * 64;
  const int ty = threadIdx.x >> 4, tx = threadIdx.x & 15;
  const bool has_wo = d.wc != nullptr;
#pragma unroll
  for (int cc = 0; cc < 4; ++cc) {
    const int cl = ty * 4 + cc, c = c0 + cl, r = r0 + tx * 4;
    float o[4] = {0.f, 0.f, 0.f, 0.f};
    if (c < d.col && r < d.row) {  // row % 4 == 0 so the float4 is entirely in range
      const float4 w = *(const float4*)(d.W + (size_t)c * d.row + r);
      o[0] = w.x; o[1] = w.y; o[2] = w.z; o[3] = w.w;
      if (has_wo) {
        const float bcv = V_b(d)[c], scv = V_s(d)[c], brv = d.br[c];
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          const float off = V_a(d)[r + j] * bcv + d.bc[r + j] * scv + brv;
          o[j] = (d.mode & E4T_WO_OFFSETS_ONLY) ? off : o[j] * (1.f + off);
        }
      }
      if (d.weff) {
        if (d.mode & E4T_WO_STORE_F32) *(float4*)((float*)d.weff + (size_t)c * d.ld_weff + r) = make_float4(o[0], o[1], o[2], o[3]);
        else *(uint2*)((bf16_t*)d.weff + (size_t)c * d.ld_weff + r) = pack4(o);
      }
    }
#pragma unroll
    for (int j = 0; j < 4; ++j) tile[cl][tx * 4 + j] = f2bf(o[j]);
  }
  __syncthreads();
  if (d.weffT) {
#pragma unroll
    for (int rr = 0; rr < 4; ++rr) {
      const int rl = ty * 4 + rr, r = r0 + rl, c = c0 + tx * 4;
      if (r < d.row && c < d.col) {  // col % 4 == 0
        uint2 o;
        o.x = (uint32_t)tile[tx * 4 + 0][rl] | ((uint32_t)tile[tx * 4 + 1][rl] << 16);
        o.y = (uint32_t)tile[tx * 4 + 2][rl] | ((uint32_t)tile[tx * 4 + 3][rl] << 16);
        *(uint2*)((bf16_t*)d.weffT + (size_t)r * d.ld_weffT + c) = o;
      }
    }
  }
}

// ---- backward 1 (one wave per c): dbr, db, ds ; optional dW ----
__global__ __launch_bounds__(256) void wo_bwd_row_kernel(const e4t_wo_desc* descs, int accum) {
  const e4t_wo_desc d = descs[blockIdx.y];
  const int c = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
  if (c >= d.col) return;
  const float* a = V_a(d);
  const float bcv = V_b(d)[c], scv = V_s(d)[c], brv = d.br[c];
  float s0 = 0.f, s1 = 0.f, s2 = 0.f;
  for (int r = lane * 4; r < d.row; r += 256) {
    const float4 g4 = *(const float4*)(d.dweff + (size_t)c * d.ld_dweff + r);
    const float4 w4 = *(const float4*)(d.W + (size_t)c * d.row + r);
    const float4 a4 = *(const float4*)(a + r), bc4 = *(const float4*)(d.bc + r);
    const float g[4] = {g4.x * w4.x, g4.y * w4.y, g4.z * w4.z, g4.w * w4.w};
    const float av[4] = {a4.x, a4.y, a4.z, a4.w}, bv[4] = {bc4.x, bc4.y, bc4.z, bc4.w};
    const float dg[4] = {g4.x, g4.y, g4.z, g4.w};
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      s0 += g[j]; s1 += g[j] * av[j]; s2 += g[j] * bv[j];
      if (d.g_W) {
        float* p = d.g_W + (size_t)c * d.row + r + j;
        *p = accw(p, dg[j] * (1.f + av[j] * bcv + bv[j] * scv + brv), accum);
      }
    }
  }
  s0 = wave_sum(s0); s1 = wave_sum(s1); s2 = wave_sum(s2);
  if (lane == 0) {
    d.g_br[c] = accw(d.g_br + c, s0, accum);
    V_db(d)[c] = s1; V_ds(d)[c] = s2;
  }
}

// ---- backward 2: column partials over 32-row chunks of c: partial[cb][r] = (sum g*s[c], sum g*b[c]) ----
__global__ __launch_bounds__(256) void wo_bwd_col_kernel(const e4t_wo_desc* descs) {
  const e4t_wo_desc d = descs[blockIdx.y];
  const int nrb = (d.row + 255) >> 8, ncb = (d.col + 31) >> 5;
  if ((int)blockIdx.x >= nrb * ncb) return;
  const int cb = blockIdx.x / nrb, r = (blockIdx.x % nrb) * 256 + threadIdx.x;
  if (r >= d.row) return;
  const float* b = V_b(d); const float* s = V_s(d);
  int c1 = cb * 32 + 32; if (c1 > d.col) c1 = d.col;
  float p0 = 0.f, p1 = 0.f;
  for (int c = cb * 32; c < c1; ++c) {
    const float g = d.dweff[(size_t)c * d.ld_dweff + r] * d.W[(size_t)c * d.row + r];
    p0 += g * s[c]; p1 += g * b[c];
  }
  float* o = d.partial + ((size_t)cb * d.row + r) * 2;
  o[0] = p0; o[1] = p1;
}

// ---- backward 3: finalize dbc, da ----
__global__ __launch_bounds__(256) void wo_bwd_colfin_kernel(const e4t_wo_desc* descs, int accum) {
  const e4t_wo_desc d = descs[blockIdx.y];
  const int r = blockIdx.x * 256 + threadIdx.x;
  if (r >= d.row) return;
  const int ncb = (d.col + 31) >> 5;
  float p0 = 0.f, p1 = 0.f;
  for (int cb = 0; cb < ncb; ++cb) {
    const float* o = d.partial + ((size_t)cb * d.row + r) * 2;
    p0 += o[0]; p1 += o[1];
  }
  d.g_bc[r] = accw(d.g_bc + r, p0, accum);
  V_da(d)[r] = p1;
}

// ---- backward 4: dWc = da (x) vx ; dWr = db (x) vy + ds ; partial dvx / dvy over 32-row chunks ----
// partial2 region (after the [ncb][row][2] region): pvx[nrc][row] then pvy[ncb][col]
__global__ __launch_bounds__(256) void wo_bwd_outer_kernel(const e4t_wo_desc* descs, int accum) {
  const e4t_wo_desc d = descs[blockIdx.y];
  const int ncb = (d.col + 31) >> 5, nrc = (d.row + 31) >> 5;
  const int nkr = (d.row + 255) >> 8, nkc = (d.col + 255) >> 8;
  const int n_wc = nrc * nkr, n_wr = ncb * nkc;
  float* pvx = d.partial + (size_t)ncb * d.row * 2;
  float* pvy = pvx + (size_t)nrc * d.row;
  int bx = blockIdx.x;
  if (bx < n_wc) {
    const int rc = bx / nkr, k = (bx % nkr) * 256 + threadIdx.x;
    if (k >= d.row) return;
    const float vxk = V_vx(d)[k];
    const float* da = V_da(d);
    int r1 = rc * 32 + 32; if (r1 > d.row) r1 = d.row;
    float p = 0.f;
    for (int r = rc * 32; r < r1; ++r) {
      const size_t idx = (size_t)r * d.row + k;
      p += d.wc[idx] * da[r];
      d.g_wc[idx] = accw(d.g_wc + idx, da[r] * vxk, accum);
    }
    pvx[(size_t)rc * d.row + k] = p;
    return;
  }
  bx -= n_wc;
  if (bx < n_wr) {
    const int cb = bx / nkc, k = (bx % nkc) * 256 + threadIdx.x;
    if (k >= d.col) return;
    const float vyk = V_vy(d)[k];
    const float* db = V_db(d); const float* ds = V_ds(d);
    int c1 = cb * 32 + 32; if (c1 > d.col) c1 = d.col;
    float p = 0.f;
    for (int c = cb * 32; c < c1; ++c) {
      const size_t idx = (size_t)c * d.col + k;
      p += d.wr[idx] * db[c];
      d.g_wr[idx] = accw(d.g_wr + idx, db[c] * vyk + ds[c], accum);
    }
    pvy[(size_t)cb * d.col + k] = p;
  }
}

// ---- backward 5: dvx, dvy -> dw1, db1, dw2, db2, dv (one block per instance) ----
__global__ __launch_bounds__(256) void wo_bwd_final_kernel(const e4t_wo_desc* descs, int accum) {
  const e4t_wo_desc d = descs[blockIdx.x];
  __shared__ float red[16];
  const int ncb = (d.col + 31) >> 5, nrc = (d.row + 31) >> 5;
  const float* pvx = d.partial + (size_t)ncb * d.row * 2;
  const float* pvy = pvx + (size_t)nrc * d.row;
  const float v = d.v[0];
  float dv = 0.f;
  for (int k = threadIdx.x; k < d.row; k += 256) {
    float t = 0.f;
    for (int rc = 0; rc < nrc; ++rc) t += pvx[(size_t)rc * d.row + k];
    d.g_w1[k] = accw(d.g_w1 + k, t * v, accum);
    d.g_b1[k] = accw(d.g_b1 + k, t, accum);
    dv += t * d.w1[k];
  }
  for (int k = threadIdx.x; k < d.col; k += 256) {
    float t = 0.f;
    for (int cb = 0; cb < ncb; ++cb) t += pvy[(size_t)cb * d.col + k];
    d.g_w2[k] = accw(d.g_w2 + k, t * v, accum);
    d.g_b2[k] = accw(d.g_b2 + k, t, accum);
    dv += t * d.w2[k];
  }
  dv = block_sum(dv, red);
  if (threadIdx.x == 0) d.g_v[0] = accw(d.g_v, dv, accum);
}

}  // namespace

extern "C" size_t e4t_wo_vecs_floats(int row, int col) { return (size_t)4 * row + (size_t)5 * col; }
extern "C" size_t e4t_wo_partial_floats(int row, int col) {
  const size_t ncb = (col + 31) / 32, nrc = (row + 31) / 32;
  return ncb * row * 2 + nrc * row + ncb * col;
}

extern "C" int e4t_wo_forward(const e4t_wo_desc* descs_dev, int n, int max_row, int max_col, e4t_stream stream) {
  E4T_REQUIRE(descs_dev && n > 0 && max_row > 0 && max_col > 0, "wo_forward: bad arguments");
  E4T_REQUIRE(max_row % 4 == 0 && max_col % 4 == 0, "wo_forward: dims must be multiples of 4");
  hipStream_t st = (hipStream_t)stream;
  hipLaunchKernelGGL(wo_vec_kernel, dim3(cdiv(max_row + max_col, 4), n), dim3(256), 0, st, descs_dev);
  E4T_CHECK_LAUNCH("wo_vec_kernel");
  hipLaunchKernelGGL(wo_apply_kernel, dim3(cdiv(max_row, 64) * cdiv(max_col, 64), n), dim3(256), 0, st, descs_dev);
  E4T_CHECK_LAUNCH("wo_apply_kernel");
  return 0;
}

// cast-only variant for plain weights (wc == nullptr in every descriptor): skips the vector pass
extern "C" int e4t_weight_prepare(const e4t_wo_desc* descs_dev, int n, int max_row, int max_col, e4t_stream stream) {
  E4T_REQUIRE(descs_dev && n > 0 && max_row > 0 && max_col > 0, "weight_prepare: bad arguments");
  E4T_REQUIRE(max_row % 4 == 0 && max_col % 4 == 0, "weight_prepare: dims must be multiples of 4");
  hipLaunchKernelGGL(wo_apply_kernel, dim3(cdiv(max_row, 64) * cdiv(max_col, 64), n), dim3(256), 0, (hipStream_t)stream, descs_dev);
  E4T_CHECK_LAUNCH("wo_apply_kernel");
  return 0;
}

extern "C" int e4t_wo_backward(const e4t_wo_desc* descs_dev, int n, int max_row, int max_col, int accumulate, e4t_stream stream) {
  E4T_REQUIRE(descs_dev && n > 0 && max_row > 0 && max_col > 0, "wo_backward: bad arguments");
  hipStream_t st = (hipStream_t)stream;
  hipLaunchKernelGGL(wo_bwd_row_kernel, dim3(cdiv(max_col, 4), n), dim3(256), 0, st, descs_dev, accumulate);
  E4T_CHECK_LAUNCH("wo_bwd_row_kernel");
  hipLaunchKernelGGL(wo_bwd_col_kernel, dim3(cdiv(max_row, 256) * cdiv(max_col, 32), n), dim3(256), 0, st, descs_dev);
  E4T_CHECK_LAUNCH("wo_bwd_col_kernel");
  hipLaunchKernelGGL(wo_bwd_colfin_kernel, dim3(cdiv(max_row, 256), n), dim3(256), 0, st, descs_dev, accumulate);
  E4T_CHECK_LAUNCH("wo_bwd_colfin_kernel");
  const int n_outer = cdiv(max_row, 32) * cdiv(max_row, 256) + cdiv(max_col, 32) * cdiv(max_col, 256);
  hipLaunchKernelGGL(wo_bwd_outer_kernel, dim3(n_outer, n), dim3(256), 0, st, descs_dev, accumulate);
  E4T_CHECK_LAUNCH("wo_bwd_outer_kernel");
  hipLaunchKernelGGL(wo_bwd_final_kernel, dim3(n), dim3(256), 0, st, descs_dev, accumulate);
  E4T_CHECK_LAUNCH("wo_bwd_final_kernel");
  return 0;
}
