// Streaming (HBM-bound) helper kernels on NHWC bf16 activations: GEGLU, SiLU, transposes, weight
// relayout, 2x2 sum-pool (upsample backward), spatial mean (E4T UNet-feature pooling), sinusoidal
// timestep embedding, bicubic resize + CLIP normalisation + patchify, fused AdamW.
// All use 8/16-byte vector accesses with consecutive lanes on consecutive addresses.
#include "common.h"
#include "../../include/e4t_hip.h"

namespace {

// ---- GEGLU: h[m][j] = u[m][j] * gelu(u[m][H + j]),  u = proj(x) of width 2H  (attention.py:428-430) ----
__global__ __launch_bounds__(256) void geglu_fwd_kernel(const bf16_t* u, bf16_t* h, long long M, int H) {
  const int hc = H >> 3;
  const long long total = M * hc;
  for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long long)gridDim.x * 256) {
    const long long m = i / hc; const int j = (int)(i - m * hc) * 8;
    float a[8], g[8];
    unpack8(*(const uint4*)(u + m * 2 * H + j), a);
    unpack8(*(const uint4*)(u + m * 2 * H + H + j), g);
#pragma unroll
    for (int k = 0; k < 8; ++k) a[k] *= gelu_f(g[k]);
    *(uint4*)(h + m * H + j) = pack8(a);
  }
}
// du[m][j] = dh * gelu(g) ; du[m][H+j] = dh * a * gelu'(g)
__global__ __launch_bounds__(256) void geglu_bwd_kernel(const bf16_t* u, const bf16_t* dh, bf16_t* du, long long M, int H) {
  const int hc = H >> 3;
  const long long total = M * hc;
  for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long long)gridDim.x * 256) {
    const long long m = i / hc; const int j = (int)(i - m * hc) * 8;
    float a[8], g[8], d[8], oa[8], og[8];
    unpack8(*(const uint4*)(u + m * 2 * H + j), a);
    unpack8(*(const uint4*)(u + m * 2 * H + H + j), g);
    unpack8(*(const uint4*)(dh + m * H + j), d);
#pragma unroll
    for (int k = 0; k < 8; ++k) { oa[k] = d[k] * gelu_f(g[k]); og[k] = d[k] * a[k] * dgelu_f(g[k]); }
    *(uint4*)(du + m * 2 * H + j) = pack8(oa);
    *(uint4*)(du + m * 2 * H + H + j) = pack8(og);
  }
}

// ---- unary: op 0 = silu, 1 = silu backward (dy, x) , 2 = gelu, 3 = gelu backward, 4 = leaky_relu(0.01), 5 = its backward,
//            6 = quick_gelu x*sigmoid(1.702 x) (CLIP text encoder of SD-1.x), 7 = its backward ----
__global__ __launch_bounds__(256) void unary_kernel(const bf16_t* x, const bf16_t* dy, bf16_t* y, long long n8, int op) {
  for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < n8; i += (long long)gridDim.x * 256) {
    float f[8], d[8];
    unpack8(*(const uint4*)(x + i * 8), f);
    if (dy) unpack8(*(const uint4*)(dy + i * 8), d);
#pragma unroll
    for (int k = 0; k < 8; ++k) {
      switch (op) {
        case 0: f[k] = silu_f(f[k]); break;
        case 1: f[k] = d[k] * dsilu_f(f[k]); break;
        case 2: f[k] = gelu_f(f[k]); break;
        case 3: f[k] = d[k] * dgelu_f(f[k]); break;
        case 4: f[k] = f[k] > 0.f ? f[k] : 0.01f * f[k]; break;
        case 5: f[k] = f[k] > 0.f ? d[k] : 0.01f * d[k]; break;
        case 6: f[k] = f[k] / (1.f + __expf(-1.702f * f[k])); break;
        default: { const float sg = 1.f / (1.f + __expf(-1.702f * f[k])); f[k] = d[k] * sg * (1.f + 1.702f * f[k] * (1.f - sg)); } break;
      }
    }
    *(uint4*)(y + i * 8) = pack8(f);
  }
}

// ---- out = a + b (bf16), optional fp32 b ----
__global__ __launch_bounds__(256) void add_kernel(const bf16_t* a, const bf16_t* b, bf16_t* y, long long n8) {
  for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < n8; i += (long long)gridDim.x * 256) {
    float f[8], g[8];
    unpack8(*(const uint4*)(a + i * 8), f);
    unpack8(*(const uint4*)(b + i * 8), g);
#pragma unroll
    for (int k = 0; k < 8; ++k) f[k] += g[k];
    *(uint4*)(y + i * 8) = pack8(f);
  }
}

// ---- batched 2-D transpose of bf16: in[b][R][C] (row stride ldi) -> out[b][C][R] (row stride ldo) ----
__global__ __launch_bounds__(256) void transpose_kernel(const bf16_t* in, bf16_t* out, int R, int C, int ldi, int ldo,
                                                        long long bsi, long long bso) {
  __shared__ bf16_t tile[64][66];
  const int b = blockIdx.z;
  const int r0 = blockIdx.y * 64, c0 = blockIdx.x * 64;
  const int ty = threadIdx.x >> 4, tx = threadIdx.x & 15;
  in += (long long)b * bsi; out += (long long)b * bso;
#pragma unroll
  for (int k = 0; k < 4; ++k) {
    const int rl = ty * 4 + k, r = r0 + rl, c = c0 + tx * 4;
#pragma unroll
    for (int j = 0; j < 4; ++j) tile[rl][tx * 4 + j] = (r < R && c + j < C) ? in[(long long)r * ldi + c + j] : (bf16_t)0;
  }
  __syncthreads();
#pragma unroll
  for (int k = 0; k < 4; ++k) {
    const int cl = ty * 4 + k, c = c0 + cl, r = r0 + tx * 4;
    if (c < C)
#pragma unroll
      for (int j = 0; j < 4; ++j)
        if (r + j < R) out[(long long)c * ldo + r + j] = tile[tx * 4 + j][cl];
  }
}

// ---- conv weight relayout: OIHW fp32 -> fwd [O][ky][kx][Ipad] bf16 and dgrad [I][2-ky][2-kx][Opad] bf16 ----
__global__ __launch_bounds__(256) void conv_weight_kernel(const float* w, bf16_t* wf, bf16_t* wd, int O, int I, int Ipad, int Opad) {
  const long long nf = (long long)O * 9 * Ipad, nd = (long long)I * 9 * Opad;
  for (long long idx = (long long)blockIdx.x * 256 + threadIdx.x; idx < nf + nd; idx += (long long)gridDim.x * 256) {
    if (idx < nf) {
      if (!wf) continue;
      const int i = (int)(idx % Ipad); const long long t = idx / Ipad;
      const int tap = (int)(t % 9), o = (int)(t / 9);
      wf[idx] = i < I ? f2bf(w[((long long)o * I + i) * 9 + tap]) : (bf16_t)0;
    } else {
      if (!wd) continue;
      const long long j = idx - nf;
      const int o = (int)(j % Opad); const long long t = j / Opad;
      const int tapd = (int)(t % 9), i = (int)(t / 9);
      const int tap = 8 - tapd;  // (2-ky)*3 + (2-kx)
      wd[j] = o < O ? f2bf(w[((long long)o * I + i) * 9 + tap]) : (bf16_t)0;
    }
  }
}

// ---- 2x2 sum pool: out[b][y][x][c] = sum in[b][2y+dy][2x+dx][c]  (backward of nearest x2 upsample) ----
__global__ __launch_bounds__(256) void sumpool2_kernel(const bf16_t* in, bf16_t* out, int Bn, int H, int W, int C) {
  const int c8 = C >> 3;
  const long long total = (long long)Bn * H * W * c8;
  for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long long)gridDim.x * 256) {
    const int c = (int)(i % c8) * 8; long long t = i / c8;
    const int x = (int)(t % W); t /= W; const int y = (int)(t % H); const int b = (int)(t / H);
    float acc[8] = {0, 0, 0, 0, 0, 0, 0, 0};
#pragma unroll
    for (int dy = 0; dy < 2; ++dy)
#pragma unroll
      for (int dx = 0; dx < 2; ++dx) {
        float f[8];
        unpack8(*(const uint4*)(in + (((long long)b * 2 * H + 2 * y + dy) * 2 * W + 2 * x + dx) * C + c), f);
#pragma unroll
        for (int k = 0; k < 8; ++k) acc[k] += f[k];
      }
    *(uint4*)(out + (((long long)b * H + y) * W + x) * C + c) = pack8(acc);
  }
}

// ---- spatial mean: out[b][coff + c] = mean_p x[b][p][c]  (encoder.py:147) ; grid (ceil(C/256), B) ----
// block = 1024 threads = 64 pixel groups x 16 lanes; each lane owns 4 channels of a 64-channel slab (8-byte loads,
// 128 B contiguous per pixel per 16 lanes); fixed-order LDS tree -> deterministic.
__global__ __launch_bounds__(1024) void spatial_mean_kernel(const bf16_t* x, float* out, int HW, int C, int ldo, int coff) {
  __shared__ float red[64][65];
  const int b = blockIdx.y, c0 = blockIdx.x * 64;
  const int lane = threadIdx.x & 15, pg = threadIdx.x >> 4;
  const int c = c0 + lane * 4;
  float s[4] = {0.f, 0.f, 0.f, 0.f};
  if (c < C) {
    const bf16_t* p = x + (long long)b * HW * C + c;
    for (int i = pg; i < HW; i += 64) {
      float f[4];
      unpack4(*(const uint2*)(p + (long long)i * C), f);
#pragma unroll
      for (int j = 0; j < 4; ++j) s[j] += f[j];
    }
  }
#pragma unroll
  for (int j = 0; j < 4; ++j) red[pg][lane * 4 + j] = s[j];
  __syncthreads();
  if (threadIdx.x < 64 && c0 + threadIdx.x < C) {
    float t = 0.f;
    for (int g = 0; g < 64; ++g) t += red[g][threadIdx.x];
    out[(long long)b * ldo + coff + c0 + threadIdx.x] = t / HW;
  }
}
// dx[b][p][c] = (base ? base : 0) + g[b][coff + c] / HW
__global__ __launch_bounds__(256) void spatial_mean_bwd_kernel(const float* g, const bf16_t* base, bf16_t* dx, int Bn, int HW, int C, int ldg, int coff) {
  const int c8 = C >> 3;
  const long long total = (long long)Bn * HW * c8;
  const float inv = 1.f / HW;
  for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long long)gridDim.x * 256) {
    const int c = (int)(i % c8) * 8; const long long pix = i / c8; const int b = (int)(pix / HW);
    float f[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    if (base) unpack8(*(const uint4*)(base + pix * C + c), f);
#pragma unroll
    for (int k = 0; k < 8; ++k) f[k] += g[(long long)b * ldg + coff + c + k] * inv;
    *(uint4*)(dx + pix * C + c) = pack8(f);
  }
}

// ---- sinusoidal timestep embedding [cos | sin] (flip_sin_to_cos=True, freq_shift=0), bf16 out ----
__global__ void timestep_embed_kernel(const long long* t, bf16_t* out, int Bn, int dim) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  const int half = dim >> 1;
  if (i >= Bn * half) return;
  const int b = i / half, k = i - b * half;
  const float freq = __expf(-9.210340371976184f * (float)k / (float)half);
  const float ang = (float)t[b] * freq;
  out[(long long)b * dim + k] = f2bf(cosf(ang));
  out[(long long)b * dim + half + k] = f2bf(sinf(ang));
}

// ---- bicubic (A=-0.75, align_corners=True) 512->224 + (x+1)/2 + CLIP mean/std + patchify ----
// in: NCHW fp32 (B,3,Hin,Win) in [-1,1] ; out: bf16 [B*gh*gw][Kpad] with k = (c*P + py)*P + px  (conv1 weight order)
__device__ __forceinline__ void cubic_w(float t, float* w) {
  const float A = -0.75f;
  w[0] = ((A * (t + 1.f) - 5.f * A) * (t + 1.f) + 8.f * A) * (t + 1.f) - 4.f * A;
  w[1] = ((A + 2.f) * t - (A + 3.f)) * t * t + 1.f;
  w[2] = ((A + 2.f) * (1.f - t) - (A + 3.f)) * (1.f - t) * (1.f - t) + 1.f;
  w[3] = 1.f - w[0] - w[1] - w[2];
}
__global__ __launch_bounds__(256) void clip_preprocess_kernel(const float* in, bf16_t* out, int Bn, int Hin, int Win, int S, int P, int Kpad) {
  const long long total = (long long)Bn * 3 * S * S;
  const float mean[3] = {0.48145466f, 0.4578275f, 0.40821073f};
  const float stdv[3] = {0.26862954f, 0.26130258f, 0.27577711f};
  const int g = S / P;
  const float sy = S > 1 ? (float)(Hin - 1) / (float)(S - 1) : 0.f, sx = S > 1 ? (float)(Win - 1) / (float)(S - 1) : 0.f;
  for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long long)gridDim.x * 256) {
    const int ox = (int)(i % S); long long t = i / S; const int oy = (int)(t % S); t /= S; const int c = (int)(t % 3); const int b = (int)(t / 3);
    const float fy = oy * sy, fx = ox * sx;
    const int iy = (int)floorf(fy), ix = (int)floorf(fx);
    float wy[4], wx[4];
    cubic_w(fy - iy, wy); cubic_w(fx - ix, wx);
    const float* src = in + ((long long)b * 3 + c) * Hin * Win;
    float acc = 0.f;
#pragma unroll
    for (int a = 0; a < 4; ++a) {
      int yy = iy - 1 + a; yy = yy < 0 ? 0 : (yy >= Hin ? Hin - 1 : yy);
      float r = 0.f;
#pragma unroll
      for (int bb = 0; bb < 4; ++bb) {
        int xx = ix - 1 + bb; xx = xx < 0 ? 0 : (xx >= Win ? Win - 1 : xx);
        r += wx[bb] * src[(long long)yy * Win + xx];
      }
      acc += wy[a] * r;
    }
    acc = ((acc + 1.f) * 0.5f - mean[c]) / stdv[c];
    const int py = oy / P, px = ox / P;
    const long long row = ((long long)b * g + py) * g + px;
    const int k = (c * P + (oy - py * P)) * P + (ox - px * P);
    out[row * Kpad + k] = f2bf(acc);
  }
}

// ---- fused AdamW over a flat fp32 buffer (torch.optim.AdamW semantics: decoupled weight decay) ----
// hyper != nullptr: lr, bc1, bc2_sqrt and the gradient scale are read from DEVICE memory ([4] floats) instead of the launch arguments,
// so that a captured (hipGraph) training step replays with the values of ITS step (e4t_adamw_hyper); same arithmetic either way.
__global__ __launch_bounds__(256) void adamw_kernel(float* p, const float* g, float* m, float* v, long long n, float lr, float b1, float b2,
                                                    float eps, float wd, float bc1, float bc2_sqrt, float gscale, const float* hyper) {
  if (hyper) { lr = hyper[0]; bc1 = hyper[1]; bc2_sqrt = hyper[2]; gscale = hyper[3]; }
  for (long long i = ((long long)blockIdx.x * 256 + threadIdx.x) * 4; i < n; i += (long long)gridDim.x * 1024) {
    if (i + 4 <= n) {
      float4 P = *(float4*)(p + i), M = *(float4*)(m + i), V = *(float4*)(v + i);
      const float4 G = *(const float4*)(g + i);
      float pp[4] = {P.x, P.y, P.z, P.w}, mm[4] = {M.x, M.y, M.z, M.w}, vv[4] = {V.x, V.y, V.z, V.w};
      const float gg[4] = {G.x * gscale, G.y * gscale, G.z * gscale, G.w * gscale};
#pragma unroll
      for (int k = 0; k < 4; ++k) {
        pp[k] *= 1.f - lr * wd;
        mm[k] = b1 * mm[k] + (1.f - b1) * gg[k];
        vv[k] = b2 * vv[k] + (1.f - b2) * gg[k] * gg[k];
        const float denom = sqrtf(vv[k]) / bc2_sqrt + eps;
        pp[k] -= (lr / bc1) * mm[k] / denom;
      }
      *(float4*)(p + i) = make_float4(pp[0], pp[1], pp[2], pp[3]);
      *(float4*)(m + i) = make_float4(mm[0], mm[1], mm[2], mm[3]);
      *(float4*)(v + i) = make_float4(vv[0], vv[1], vv[2], vv[3]);
    } else {
      for (long long j = i; j < n; ++j) {
        float pp = p[j] * (1.f - lr * wd);
        const float gg = g[j] * gscale;
        const float mm = b1 * m[j] + (1.f - b1) * gg, vv = b2 * v[j] + (1.f - b2) * gg * gg;
        pp -= (lr / bc1) * mm / (sqrtf(vv) / bc2_sqrt + eps);
        p[j] = pp; m[j] = mm; v[j] = vv;
      }
    }
  }
}

// ---- AdamW of the E4T head's stacked first_linears weights with the gradient formed IN REGISTERS (round 6) ----
// The n = 129 weight gradients are rank-K products dW_i[r][c] = sum_k G[k][r] * Z[k][i*cols + c] (K = images of the step: the batch, or
// under data parallelism every rank's rows gathered; encoder.py:159-162): 845 MB of fp32 that the step used to write (batched GEMM), read
// back (AdamW) and clear (zero_grad).  Here a workgroup owns RB rows of one slot: a thread holds its 4 columns of the K x cols factor
// slice (bf16 -> fp32 registers, re-read from L2 by the cols / RB workgroups of the slot), the RB x K factor block of G sits in LDS
// and is read as broadcasts, and each element's gradient is a k-ordered fmaf chain in front of the same update arithmetic as
// adamw_kernel: 24 B of HBM traffic per parameter instead of 36, ~2 K FMAs per element on a kernel that stays HBM-bound.
template <int RB>
__global__ __launch_bounds__(320) void adamw_rank_kernel(float* p, float* m, float* v, const bf16_t* G, const bf16_t* Z, int rows, int cols, int K,
                                                         int ldg, long long ldz, float lr, float b1, float b2, float eps, float wd, float bc1,
                                                         float bc2_sqrt, float gscale, const float* hyper) {
  if (hyper) { lr = hyper[0]; bc1 = hyper[1]; bc2_sqrt = hyper[2]; gscale = hyper[3]; }
  __shared__ float gs[16 * RB];                      // G[k0 + k][r0 + rr] as fp32, [rr][k]
  const int slot = blockIdx.y, r0 = blockIdx.x * RB;
  const int Q = cols >> 2;
  for (int q = threadIdx.x; q < (Q + (int)blockDim.x - 1) / (int)blockDim.x * (int)blockDim.x; q += blockDim.x) {      // (uniform trip count: barriers inside)
    const bool on = q < Q;
    float acc[RB][4];
#pragma unroll
    for (int rr = 0; rr < RB; ++rr)
#pragma unroll
      for (int j = 0; j < 4; ++j) acc[rr][j] = 0.f;
    for (int k0 = 0; k0 < K; k0 += 16) {
      __syncthreads();
      for (int i = threadIdx.x; i < 16 * RB; i += blockDim.x) {
        const int rr = i >> 4, k = i & 15;
        gs[i] = (k0 + k < K && r0 + rr < rows) ? bf2f(G[(long long)(k0 + k) * ldg + r0 + rr]) : 0.f;
      }
      float z[16][4];
#pragma unroll
      for (int k = 0; k < 16; ++k) {
        uint2 w = make_uint2(0u, 0u);
        if (on && k0 + k < K) w = *(const uint2*)(Z + (long long)(k0 + k) * ldz + (long long)slot * cols + 4 * q);
        unpack4(w, z[k]);
      }
      __syncthreads();
#pragma unroll
      for (int rr = 0; rr < RB; ++rr) {
        const float4* g4 = (const float4*)(gs + rr * 16);
#pragma unroll
        for (int kk = 0; kk < 4; ++kk) {
          const float4 g = g4[kk];                   // broadcast read
          const float gk[4] = {g.x, g.y, g.z, g.w};
#pragma unroll
          for (int u = 0; u < 4; ++u)
#pragma unroll
            for (int j = 0; j < 4; ++j) acc[rr][j] = fmaf(gk[u], z[kk * 4 + u][j], acc[rr][j]);
        }
      }
    }
    if (!on) continue;
#pragma unroll
    for (int rr = 0; rr < RB; ++rr) {
      if (r0 + rr >= rows) continue;
      const long long i = ((long long)slot * rows + r0 + rr) * cols + 4 * q;
      float4 P = *(float4*)(p + i), M = *(float4*)(m + i), V = *(float4*)(v + i);
      float pp[4] = {P.x, P.y, P.z, P.w}, mm[4] = {M.x, M.y, M.z, M.w}, vv[4] = {V.x, V.y, V.z, V.w};
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const float gg = acc[rr][j] * gscale;
        pp[j] *= 1.f - lr * wd;
        mm[j] = b1 * mm[j] + (1.f - b1) * gg;
        vv[j] = b2 * vv[j] + (1.f - b2) * gg * gg;
        const float denom = sqrtf(vv[j]) / bc2_sqrt + eps;
        pp[j] -= (lr / bc1) * mm[j] / denom;
      }
      *(float4*)(p + i) = make_float4(pp[0], pp[1], pp[2], pp[3]);
      *(float4*)(m + i) = make_float4(mm[0], mm[1], mm[2], mm[3]);
      *(float4*)(v + i) = make_float4(vv[0], vv[1], vv[2], vv[3]);
    }
  }
}

// sum of squares of a flat fp32 buffer -> partial[blockIdx.x]
__global__ __launch_bounds__(256) void sumsq_kernel(const float* g, long long n, float* partial) {
  __shared__ float red[16];
  float s = 0.f;
  for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long long)gridDim.x * 256) s += g[i] * g[i];
  s = block_sum(s, red);
  if (threadIdx.x == 0) partial[blockIdx.x] = s;
}

int grid_for(long long work, int cap = 2048) {
  long long b = (work + 255) / 256;
  if (b < 1) b = 1;
  return (int)(b > cap ? cap : b);
}

}  // namespace

extern "C" int e4t_geglu_fwd(const void* u, void* h, long long M, int H, e4t_stream s) {
  E4T_REQUIRE(u && h && M > 0 && H > 0 && H % 8 == 0, "geglu_fwd: bad arguments");
  E4T_LOG_LAUNCH("geglu_fwd_kernel|M%lld H%d|%.0f|0", M, H, 6.0 * (double)M * H);
  hipLaunchKernelGGL(geglu_fwd_kernel, dim3(grid_for(M * (H / 8))), dim3(256), 0, (hipStream_t)s, (const bf16_t*)u, (bf16_t*)h, M, H);
  E4T_CHECK_LAUNCH("geglu_fwd_kernel");
  return 0;
}
extern "C" int e4t_geglu_bwd(const void* u, const void* dh, void* du, long long M, int H, e4t_stream s) {
  E4T_REQUIRE(u && dh && du && M > 0 && H > 0 && H % 8 == 0, "geglu_bwd: bad arguments");
  E4T_LOG_LAUNCH("geglu_bwd_kernel|M%lld H%d|%.0f|0", M, H, 10.0 * (double)M * H);
  hipLaunchKernelGGL(geglu_bwd_kernel, dim3(grid_for(M * (H / 8))), dim3(256), 0, (hipStream_t)s, (const bf16_t*)u, (const bf16_t*)dh, (bf16_t*)du, M, H);
  E4T_CHECK_LAUNCH("geglu_bwd_kernel");
  return 0;
}
extern "C" int e4t_unary(const void* x, const void* dy, void* y, long long n, int op, e4t_stream s) {
  E4T_REQUIRE(x && y && n > 0 && n % 8 == 0 && op >= 0 && op <= 7, "unary: bad arguments (n %% 8 == 0 required)");
  E4T_REQUIRE(((op & 1) == 0) || dy, "unary: backward ops need dy");
  hipLaunchKernelGGL(unary_kernel, dim3(grid_for(n / 8)), dim3(256), 0, (hipStream_t)s, (const bf16_t*)x, (const bf16_t*)((op & 1) ? dy : nullptr), (bf16_t*)y, n / 8, op);
  E4T_CHECK_LAUNCH("unary_kernel");
  return 0;
}
extern "C" int e4t_add(const void* a, const void* b, void* y, long long n, e4t_stream s) {
  E4T_REQUIRE(a && b && y && n > 0 && n % 8 == 0, "add: bad arguments");
  hipLaunchKernelGGL(add_kernel, dim3(grid_for(n / 8)), dim3(256), 0, (hipStream_t)s, (const bf16_t*)a, (const bf16_t*)b, (bf16_t*)y, n / 8);
  E4T_CHECK_LAUNCH("add_kernel");
  return 0;
}
// Classifier-free guidance + one linear scheduler update in a single pass (pipeline_stable_diffusion_e4t.py:209-214):
//   eps = cfg ? u + g*(c - u) : pred ;  out = c_sample*sample + c_pred*eps (+ c_noise*noise)
// coef = {g, c_sample, c_pred, c_noise} lives in device memory so a captured hipGraph of the step can be replayed with
// new coefficients.  pred may be given in the UNet's NHWC output layout ([B][HW][C]); sample / noise / out are NCHW.
__global__ __launch_bounds__(256) void guided_step_kernel(const float* __restrict__ pred, const float* __restrict__ sample, const float* __restrict__ noise,
                                                          float* __restrict__ out, const float* __restrict__ coef, int B, int C, int HW, int cfg, int nhwc) {
  const long long i = (long long)blockIdx.x * 256 + threadIdx.x;
  const long long n = (long long)B * C * HW;
  if (i >= n) return;
  const float g = coef[0], cs = coef[1], cp = coef[2], cn = coef[3];
  long long j = i;
  if (nhwc) {
    const int p = (int)(i % HW), c = (int)((i / HW) % C), b = (int)(i / ((long long)HW * C));
    j = ((long long)b * HW + p) * C + c;
  }
  float e = pred[j];
  if (cfg) {
    const float t = pred[j + n];
    e = e + g * (t - e);
  }
  float o = cs * sample[i] + cp * e;
  if (noise) o += cn * noise[i];
  out[i] = o;
}
extern "C" int e4t_guided_step(const float* pred, const float* sample, const float* noise, float* out, const float* coef,
                               int B, int C, int HW, int cfg, int pred_nhwc, e4t_stream s) {
  E4T_REQUIRE(pred && sample && out && coef && B > 0 && C > 0 && HW > 0, "guided_step: bad arguments");
  const long long n = (long long)B * C * HW;
  hipLaunchKernelGGL(guided_step_kernel, dim3((unsigned)cdivl(n, 256)), dim3(256), 0, (hipStream_t)s, pred, sample, noise, out, coef, B, C, HW, cfg, pred_nhwc);
  E4T_CHECK_LAUNCH("guided_step_kernel");
  return 0;
}
extern "C" int e4t_transpose(const void* in, void* out, int batch, int R, int C, int ldi, int ldo, long long bsi, long long bso, e4t_stream s) {
  E4T_REQUIRE(in && out && batch > 0 && R > 0 && C > 0 && ldi >= C && ldo >= R, "transpose: bad arguments");
  hipLaunchKernelGGL(transpose_kernel, dim3(cdiv(C, 64), cdiv(R, 64), batch), dim3(256), 0, (hipStream_t)s, (const bf16_t*)in, (bf16_t*)out, R, C, ldi, ldo, bsi, bso);
  E4T_CHECK_LAUNCH("transpose_kernel");
  return 0;
}
extern "C" int e4t_conv_weight_prepare(const float* w_oihw, void* w_fwd, void* w_dgrad, int O, int I, int Ipad, int Opad, e4t_stream s) {
  E4T_REQUIRE(w_oihw && (w_fwd || w_dgrad) && O > 0 && I > 0 && Ipad >= I && Opad >= O, "conv_weight_prepare: bad arguments");
  const long long n = (long long)O * 9 * Ipad + (long long)I * 9 * Opad;
  hipLaunchKernelGGL(conv_weight_kernel, dim3(grid_for(n, 4096)), dim3(256), 0, (hipStream_t)s, w_oihw, (bf16_t*)w_fwd, (bf16_t*)w_dgrad, O, I, Ipad, Opad);
  E4T_CHECK_LAUNCH("conv_weight_kernel");
  return 0;
}
extern "C" int e4t_sumpool2(const void* in, void* out, int Bn, int H, int W, int C, e4t_stream s) {
  E4T_REQUIRE(in && out && Bn > 0 && H > 0 && W > 0 && C % 8 == 0, "sumpool2: bad arguments");
  hipLaunchKernelGGL(sumpool2_kernel, dim3(grid_for((long long)Bn * H * W * (C / 8))), dim3(256), 0, (hipStream_t)s, (const bf16_t*)in, (bf16_t*)out, Bn, H, W, C);
  E4T_CHECK_LAUNCH("sumpool2_kernel");
  return 0;
}
extern "C" int e4t_spatial_mean(const void* x, float* out, int Bn, int HW, int C, int ldo, int coff, e4t_stream s) {
  E4T_REQUIRE(x && out && Bn > 0 && HW > 0 && C > 0 && C % 4 == 0, "spatial_mean: bad arguments (C %% 4 == 0)");
  hipLaunchKernelGGL(spatial_mean_kernel, dim3(cdiv(C, 64), Bn), dim3(1024), 0, (hipStream_t)s, (const bf16_t*)x, out, HW, C, ldo, coff);
  E4T_CHECK_LAUNCH("spatial_mean_kernel");
  return 0;
}
extern "C" int e4t_spatial_mean_bwd(const float* g, const void* base, void* dx, int Bn, int HW, int C, int ldg, int coff, e4t_stream s) {
  E4T_REQUIRE(g && dx && Bn > 0 && HW > 0 && C % 8 == 0, "spatial_mean_bwd: bad arguments");
  hipLaunchKernelGGL(spatial_mean_bwd_kernel, dim3(grid_for((long long)Bn * HW * (C / 8))), dim3(256), 0, (hipStream_t)s, g, (const bf16_t*)base, (bf16_t*)dx, Bn, HW, C, ldg, coff);
  E4T_CHECK_LAUNCH("spatial_mean_bwd_kernel");
  return 0;
}
extern "C" int e4t_timestep_embedding(const long long* t, void* out, int Bn, int dim, e4t_stream s) {
  E4T_REQUIRE(t && out && Bn > 0 && dim > 0 && dim % 2 == 0, "timestep_embedding: bad arguments");
  hipLaunchKernelGGL(timestep_embed_kernel, dim3(cdiv(Bn * dim / 2, 256)), dim3(256), 0, (hipStream_t)s, t, (bf16_t*)out, Bn, dim);
  E4T_CHECK_LAUNCH("timestep_embed_kernel");
  return 0;
}
extern "C" int e4t_clip_preprocess(const float* pixels_nchw, void* patches, int Bn, int Hin, int Win, int S, int P, int Kpad, e4t_stream s) {
  E4T_REQUIRE(pixels_nchw && patches && Bn > 0 && S % P == 0 && Kpad >= 3 * P * P, "clip_preprocess: bad arguments");
  hipLaunchKernelGGL(clip_preprocess_kernel, dim3(grid_for((long long)Bn * 3 * S * S)), dim3(256), 0, (hipStream_t)s, pixels_nchw, (bf16_t*)patches, Bn, Hin, Win, S, P, Kpad);
  E4T_CHECK_LAUNCH("clip_preprocess_kernel");
  return 0;
}
extern "C" int e4t_adamw(float* p, const float* g, float* m, float* v, long long n, float lr, float beta1, float beta2, float eps,
                         float weight_decay, int step, float grad_scale, e4t_stream s) {
  E4T_REQUIRE(p && g && m && v && n > 0 && step >= 1, "adamw: bad arguments");
  E4T_REQUIRE((((uintptr_t)p | (uintptr_t)g | (uintptr_t)m | (uintptr_t)v) & 15) == 0, "adamw: buffers must be 16-B aligned");
  const float bc1 = 1.f - powf(beta1, (float)step), bc2 = sqrtf(1.f - powf(beta2, (float)step));
  E4T_LOG_LAUNCH("adamw_kernel|n%lld|%.0f|0", (long long)n, 28.0 * (double)n);
  hipLaunchKernelGGL(adamw_kernel, dim3(grid_for((n + 3) / 4, 4096)), dim3(256), 0, (hipStream_t)s, p, g, m, v, n, lr, beta1, beta2, eps, weight_decay, bc1, bc2, grad_scale,
                     (const float*)nullptr);
  E4T_CHECK_LAUNCH("adamw_kernel");
  return 0;
}
extern "C" int e4t_adamw_hyper(float* p, const float* g, float* m, float* v, long long n, const float* hyper_dev, float beta1, float beta2, float eps,
                               float weight_decay, e4t_stream s) {
  E4T_REQUIRE(p && g && m && v && hyper_dev && n > 0, "adamw_hyper: bad arguments");
  E4T_REQUIRE((((uintptr_t)p | (uintptr_t)g | (uintptr_t)m | (uintptr_t)v | (uintptr_t)hyper_dev) & 15) == 0, "adamw_hyper: buffers must be 16-B aligned");
  E4T_LOG_LAUNCH("adamw_kernel|n%lld|%.0f|0", (long long)n, 28.0 * (double)n);
  hipLaunchKernelGGL(adamw_kernel, dim3(grid_for((n + 3) / 4, 4096)), dim3(256), 0, (hipStream_t)s, p, g, m, v, n, 0.f, beta1, beta2, eps, weight_decay, 1.f, 1.f, 1.f, hyper_dev);
  E4T_CHECK_LAUNCH("adamw_kernel");
  return 0;
}
extern "C" int e4t_adamw_rank(float* p, float* m, float* v, const void* G, const void* Z, int n, int rows, int cols, int K, int ldg, long long ldz,
                              float lr, float beta1, float beta2, float eps, float weight_decay, int step, float grad_scale, const float* hyper_dev,
                              e4t_stream s) {
  E4T_REQUIRE(p && m && v && G && Z && n > 0 && rows > 0 && cols > 0 && K > 0 && (hyper_dev || step >= 1), "adamw_rank: bad arguments");
  E4T_REQUIRE(cols % 4 == 0 && ldz % 4 == 0 && ldg >= rows && ldz >= (long long)n * cols, "adamw_rank: cols and the Z row stride must be multiples of 4");
  E4T_REQUIRE((((uintptr_t)p | (uintptr_t)m | (uintptr_t)v) & 15) == 0 && ((uintptr_t)Z & 7) == 0 && ((uintptr_t)hyper_dev & 15) == 0,
              "adamw_rank: buffers must be 16-B (factors 8-B) aligned");
  const float bc1 = hyper_dev ? 1.f : 1.f - powf(beta1, (float)step), bc2 = hyper_dev ? 1.f : sqrtf(1.f - powf(beta2, (float)step));
  constexpr int RB = 8;
  const int threads = min(320, (cols / 4 + 63) / 64 * 64);
  // algorithmic bytes: p, m, v read and written; the factors are L2 traffic
  E4T_LOG_LAUNCH("adamw_rank_kernel<%d>|n%d rows%d cols%d K%d|%.0f|%.0f", RB, n, rows, cols, K, 24.0 * n * rows * cols, 2.0 * n * rows * (double)cols * K);
  hipLaunchKernelGGL((adamw_rank_kernel<RB>), dim3(cdiv(rows, RB), n), dim3(threads), 0, (hipStream_t)s, p, m, v, (const bf16_t*)G, (const bf16_t*)Z, rows, cols, K,
                     ldg, ldz, hyper_dev ? 0.f : lr, beta1, beta2, eps, weight_decay, bc1, bc2, hyper_dev ? 1.f : grad_scale, hyper_dev);
  E4T_CHECK_LAUNCH("adamw_rank_kernel");
  return 0;
}
extern "C" int e4t_sumsq_partial(const float* g, long long n, float* partial, int nblocks, e4t_stream s) {
  E4T_REQUIRE(g && partial && n > 0 && nblocks > 0, "sumsq_partial: bad arguments");
  hipLaunchKernelGGL(sumsq_kernel, dim3(nblocks), dim3(256), 0, (hipStream_t)s, g, n, partial);
  E4T_CHECK_LAUNCH("sumsq_kernel");
  return 0;
}

// ---- in-place row softmax (VAE mid-block attention scores; fp32 math, one 256-thread block per row) ----
namespace {
__global__ __launch_bounds__(256) void softmax_rows_kernel(bf16_t* x, int L, int ld) {
  __shared__ float red[16];
  bf16_t* row = x + (long long)blockIdx.x * ld;
  const int nch = L >> 3;
  float v[8][8];
  float mx = -INFINITY;
#pragma unroll
  for (int i = 0; i < 8; ++i) {
    const int c = threadIdx.x + i * 256;
    if (c < nch) {
      unpack8(*(const uint4*)(row + c * 8), v[i]);
#pragma unroll
      for (int j = 0; j < 8; ++j) mx = fmaxf(mx, v[i][j]);
    }
  }
  mx = wave_max(mx);
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = mx;
  __syncthreads();
  mx = fmaxf(fmaxf(red[0], red[1]), fmaxf(red[2], red[3]));
  float s = 0.f;
#pragma unroll
  for (int i = 0; i < 8; ++i) {
    const int c = threadIdx.x + i * 256;
    if (c < nch)
#pragma unroll
      for (int j = 0; j < 8; ++j) { v[i][j] = __expf(v[i][j] - mx); s += v[i][j]; }
  }
  s = block_sum(s, red + 8);
  const float inv = 1.f / s;
#pragma unroll
  for (int i = 0; i < 8; ++i) {
    const int c = threadIdx.x + i * 256;
    if (c < nch) {
#pragma unroll
      for (int j = 0; j < 8; ++j) v[i][j] *= inv;
      *(uint4*)(row + c * 8) = pack8(v[i]);
    }
  }
}

__global__ __launch_bounds__(256) void im2col3_kernel(const float* px, bf16_t* out, int Bn, int H, int W) {
  const long long total = (long long)Bn * H * W;
  for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long long)gridDim.x * 256) {
    const int x = (int)(i % W); long long t = i / W; const int y = (int)(t % H); const int b = (int)(t / H);
    float f[32];
#pragma unroll
    for (int k = 0; k < 32; ++k) f[k] = 0.f;
#pragma unroll
    for (int ky = 0; ky < 3; ++ky)
#pragma unroll
      for (int kx = 0; kx < 3; ++kx) {
        const int yy = y + ky - 1, xx = x + kx - 1;
        if (yy >= 0 && yy < H && xx >= 0 && xx < W)
#pragma unroll
          for (int c = 0; c < 3; ++c) f[(ky * 3 + kx) * 3 + c] = px[(((long long)b * 3 + c) * H + yy) * W + xx];
      }
#pragma unroll
    for (int k = 0; k < 4; ++k) *(uint4*)(out + i * 32 + k * 8) = pack8(f + k * 8);
  }
}
}  // namespace

extern "C" int e4t_softmax_rows(void* x, long long rows, int L, int ld, e4t_stream s) {
  E4T_REQUIRE(x && rows > 0 && L > 0 && L % 8 == 0 && L <= 16384 && ld >= L && ld % 8 == 0, "softmax_rows: bad arguments");
  E4T_REQUIRE(rows <= 0x7fffffffLL, "softmax_rows: too many rows");
  hipLaunchKernelGGL(softmax_rows_kernel, dim3((unsigned)rows), dim3(256), 0, (hipStream_t)s, (bf16_t*)x, L, ld);
  E4T_CHECK_LAUNCH("softmax_rows_kernel");
  return 0;
}
extern "C" int e4t_im2col3_rgb(const float* pixels_nchw, void* out, int Bn, int H, int W, e4t_stream s) {
  E4T_REQUIRE(pixels_nchw && out && Bn > 0 && H > 0 && W > 0, "im2col3_rgb: bad arguments");
  long long blocks = ((long long)Bn * H * W + 255) / 256;
  if (blocks > 8192) blocks = 8192;
  hipLaunchKernelGGL(im2col3_kernel, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)s, pixels_nchw, (bf16_t*)out, Bn, H, W);
  E4T_CHECK_LAUNCH("im2col3_kernel");
  return 0;
}

// ---- im2col-transpose for 3x3 conv weight gradients (tuning mode: every UNet conv weight trains) ----
// out[(tap*C + c)][m] = X[b][iy][ix][c] (zero outside), m = (b, oy, ox) over the conv's OUTPUT pixels, row stride ldo.
// With dY^T this turns dW[co][tap][ci] = sum_m dY[m][co] * X[src(m,tap)][ci] into one NT GEMM whose contraction
// runs over the pixels.  mode: E4T_CONV_S1 / _S2 / _UP2 (same gather rules as the forward conv).
namespace {
__global__ __launch_bounds__(256) void im2col_T_kernel(const bf16_t* x, bf16_t* out, int Hin, int Win, int C, int Hout, int Wout,
                                                       long long Mpix, int ldo, int mode) {
  __shared__ bf16_t tile[64][66];
  const int tap = blockIdx.z, ky = tap / 3, kx = tap - ky * 3;
  const long long m0 = (long long)blockIdx.y * 64;
  const int c0 = blockIdx.x * 64;
  const int ty = threadIdx.x >> 4, tx = threadIdx.x & 15;
#pragma unroll
  for (int k = 0; k < 4; ++k) {
    const int rl = ty * 4 + k;
    const long long m = m0 + rl;
    bool ok = m < Mpix;
    long long src = 0;
    if (ok) {
      const int hw = Hout * Wout;
      const int b = (int)(m / hw);
      const int rem = (int)(m - (long long)b * hw);
      const int oy = rem / Wout, ox = rem - oy * Wout;
      int iy, ix;
      if (mode == E4T_CONV_S1) { iy = oy + ky - 1; ix = ox + kx - 1; ok = iy >= 0 && iy < Hin && ix >= 0 && ix < Win; }
      else if (mode == E4T_CONV_S2) { iy = 2 * oy + ky - 1; ix = 2 * ox + kx - 1; ok = iy >= 0 && iy < Hin && ix >= 0 && ix < Win; }
      else { iy = oy + ky - 1; ix = ox + kx - 1; ok = iy >= 0 && iy < 2 * Hin && ix >= 0 && ix < 2 * Win; iy >>= 1; ix >>= 1; }
      src = (((long long)b * Hin + iy) * Win + ix) * C;
    }
    const int c = c0 + tx * 4;
    uint2 v = make_uint2(0, 0);
    if (ok && c < C) v = *(const uint2*)(x + src + c);      // C % 4 == 0
    tile[rl][tx * 4 + 0] = (bf16_t)(v.x & 0xffff); tile[rl][tx * 4 + 1] = (bf16_t)(v.x >> 16);
    tile[rl][tx * 4 + 2] = (bf16_t)(v.y & 0xffff); tile[rl][tx * 4 + 3] = (bf16_t)(v.y >> 16);
  }
  __syncthreads();
#pragma unroll
  for (int k = 0; k < 4; ++k) {
    const int cl = ty * 4 + k, c = c0 + cl;
    const long long m = m0 + tx * 4;
    if (c < C && m < ldo) {                                   // ldo % 4 == 0; columns in [Mpix, ldo) receive the zero fill
      uint2 o;
      o.x = (uint32_t)tile[tx * 4 + 0][cl] | ((uint32_t)tile[tx * 4 + 1][cl] << 16);
      o.y = (uint32_t)tile[tx * 4 + 2][cl] | ((uint32_t)tile[tx * 4 + 3][cl] << 16);
      *(uint2*)(out + ((long long)tap * C + c) * ldo + m) = o;
    }
  }
}
}  // namespace

// ---- plain im2col for the same weight gradients through the TN GEMM (no transposes at all): out[m][tap*C + c] ----
namespace {
__global__ __launch_bounds__(256) void im2col_kernel(const bf16_t* x, bf16_t* out, int Hin, int Win, int C, int Hout, int Wout,
                                                     long long Mpix, int mode) {
  const int cpr = C >> 3;                                     // 16-byte chunks per (pixel, tap)
  const long long total = Mpix * 9 * cpr;
  for (long long idx = (long long)blockIdx.x * 256 + threadIdx.x; idx < total; idx += (long long)gridDim.x * 256) {
    const int ch = (int)(idx % cpr);
    const long long t = idx / cpr;
    const int tap = (int)(t % 9);
    const long long m = t / 9;
    const int ky = tap / 3, kx = tap - ky * 3;
    const int hw = Hout * Wout;
    const int b = (int)(m / hw);
    const int rem = (int)(m - (long long)b * hw);
    const int oy = rem / Wout, ox = rem - oy * Wout;
    int iy, ix;
    bool ok;
    if (mode == E4T_CONV_S1) { iy = oy + ky - 1; ix = ox + kx - 1; ok = iy >= 0 && iy < Hin && ix >= 0 && ix < Win; }
    else if (mode == E4T_CONV_S2) { iy = 2 * oy + ky - 1; ix = 2 * ox + kx - 1; ok = iy >= 0 && iy < Hin && ix >= 0 && ix < Win; }
    else { iy = oy + ky - 1; ix = ox + kx - 1; ok = iy >= 0 && iy < 2 * Hin && ix >= 0 && ix < 2 * Win; iy >>= 1; ix >>= 1; }
    uint4 v = make_uint4(0, 0, 0, 0);
    if (ok) v = *(const uint4*)(x + (((long long)b * Hin + iy) * Win + ix) * C + ch * 8);
    *(uint4*)(out + (m * 9 + tap) * C + ch * 8) = v;
  }
}
}  // namespace

extern "C" int e4t_im2col(const void* x, void* out, int Bn, int Hin, int Win, int C, int Hout, int Wout, int mode, e4t_stream s) {
  E4T_REQUIRE(x && out && Bn > 0 && Hin > 0 && Win > 0 && Hout > 0 && Wout > 0 && C % 8 == 0, "im2col: bad arguments (C must be a multiple of 8)");
  E4T_REQUIRE(mode == E4T_CONV_S1 || mode == E4T_CONV_S2 || mode == E4T_CONV_UP2, "im2col: mode must be S1, S2 or UP2");
  const long long Mpix = (long long)Bn * Hout * Wout;
  long long blocks = (Mpix * 9 * (C / 8) + 255) / 256;
  if (blocks > 65536) blocks = 65536;
  hipLaunchKernelGGL(im2col_kernel, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)s, (const bf16_t*)x, (bf16_t*)out, Hin, Win, C, Hout, Wout,
                     Mpix, mode);
  E4T_CHECK_LAUNCH("im2col_kernel");
  return 0;
}

extern "C" int e4t_im2col_T(const void* x, void* out, int Bn, int Hin, int Win, int C, int Hout, int Wout, int ldo, int mode, e4t_stream s) {
  E4T_REQUIRE(x && out && Bn > 0 && Hin > 0 && Win > 0 && Hout > 0 && Wout > 0, "im2col_T: bad arguments");
  E4T_REQUIRE(C % 4 == 0 && ldo % 4 == 0 && (long long)ldo >= (long long)Bn * Hout * Wout, "im2col_T: C, ldo must be multiples of 4, ldo >= B*Hout*Wout");
  E4T_REQUIRE(mode == E4T_CONV_S1 || mode == E4T_CONV_S2 || mode == E4T_CONV_UP2, "im2col_T: mode must be S1, S2 or UP2");
  const long long Mpix = (long long)Bn * Hout * Wout;
  hipLaunchKernelGGL(im2col_T_kernel, dim3(cdiv(C, 64), (unsigned)((ldo + 63) / 64), 9), dim3(256), 0, (hipStream_t)s, (const bf16_t*)x, (bf16_t*)out,
                     Hin, Win, C, Hout, Wout, Mpix, ldo, mode);
  E4T_CHECK_LAUNCH("im2col_T_kernel");
  return 0;
}
