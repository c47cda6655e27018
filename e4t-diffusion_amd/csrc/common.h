// Shared device/host helpers for libe4t_hip.so (gfx950 / CDNA4 only, wave64).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <string.h>

typedef uint16_t bf16_t;  // raw bfloat16 bits; all bf16 buffers cross the C ABI as untyped pointers

typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(4))) __bf16 bf16x4;
typedef __attribute__((ext_vector_type(16))) float f32x16;
typedef __attribute__((ext_vector_type(4))) float f32x4;

#define E4T_WAVE 64

// One bf16x8 MFMA operand gathered with two transposing LDS reads (ds_read_b64_tr_b16, gfx950): each lane passes the address
// of 4 contiguous bf16; within a 16-lane group, output lane i receives in slot j element (i % 4) of source lane 4j + i / 4.
typedef __attribute__((ext_vector_type(4))) short s16x4;
__device__ __forceinline__ bf16x8 tr_frag(const uint16_t* lds_lo, const uint16_t* lds_hi) {
  union { s16x4 h[2]; bf16x8 v; } u;
  u.h[0] = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4*)lds_lo);
  u.h[1] = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4*)lds_hi);
  return u.v;
}

// ---- error plumbing (never throws across the ABI) -------------------------------------------
// library-internal helpers shared by the translation units (not part of the C ABI: hidden)
extern "C" __attribute__((visibility("hidden"))) void e4t_set_error(const char* msg);
#define E4T_FAIL(code, ...)                                  \
  do {                                                       \
    char _b[512];                                            \
    snprintf(_b, sizeof(_b), __VA_ARGS__);                   \
    e4t_set_error(_b);                                       \
    return (code);                                           \
  } while (0)
#define E4T_REQUIRE(cond, ...) \
  do {                         \
    if (!(cond)) E4T_FAIL(-22, __VA_ARGS__); \
  } while (0)
#define E4T_CHECK_LAUNCH(name)                                                    \
  do {                                                                            \
    hipError_t _e = hipGetLastError();                                            \
    if (_e != hipSuccess) E4T_FAIL(-5, "%s: %s", name, hipGetErrorString(_e));    \
  } while (0)

// ---- launch log (diagnostics): one text line per kernel launch — symbol | shape | algorithmic bytes | flops — so that a
// rocprofv3 kernel trace / PMC collection of the same process can be joined per SHAPE (tools/roofline_report.py).
// Off unless E4T_LAUNCH_LOG=<path> is set (or e4t_set_launch_log() was called); one branch per launch when off.
extern "C" __attribute__((visibility("hidden"))) int e4t_launch_log_enabled(void);
extern "C" __attribute__((visibility("hidden"))) void e4t_launch_logf(const char* fmt, ...);
#define E4T_LOG_LAUNCH(...)                                          \
  do {                                                               \
    if (e4t_launch_log_enabled()) e4t_launch_logf(__VA_ARGS__);      \
  } while (0)

// ---- bf16 <-> f32 -------------------------------------------------------------------------------
__device__ __forceinline__ float bf2f(bf16_t v) { return __uint_as_float(((uint32_t)v) << 16); }
// fp32 -> bf16, round-to-nearest-even.  Written as native __bf16 conversions so that hipcc emits gfx950's
// v_cvt_pk_bf16_f32 (ONE VALU op per pair) instead of ~10 integer ops of a hand-rolled rounding.
typedef __attribute__((ext_vector_type(2))) __bf16 bf16x2_native;
typedef __attribute__((ext_vector_type(2))) float f32x2_native;
__device__ __forceinline__ bf16_t f2bf(float f) {
  const __bf16 b = (__bf16)f;
  bf16_t u;
  __builtin_memcpy(&u, &b, 2);
  return u;
}
__device__ __forceinline__ uint32_t pack2bf(float lo, float hi) {
  const f32x2_native f = {lo, hi};
  const bf16x2_native v = __builtin_convertvector(f, bf16x2_native);
  uint32_t u;
  __builtin_memcpy(&u, &v, 4);
  return u;
}
__device__ __forceinline__ float bflo(uint32_t w) { return __uint_as_float(w << 16); }
__device__ __forceinline__ float bfhi(uint32_t w) { return __uint_as_float(w & 0xffff0000u); }

// 8 bf16 (one uint4) <-> 8 floats
__device__ __forceinline__ void unpack8(const uint4& v, float* f) {
  f[0] = bflo(v.x); f[1] = bfhi(v.x); f[2] = bflo(v.y); f[3] = bfhi(v.y);
  f[4] = bflo(v.z); f[5] = bfhi(v.z); f[6] = bflo(v.w); f[7] = bfhi(v.w);
}
__device__ __forceinline__ uint4 pack8(const float* f) {
  uint4 v;
  v.x = pack2bf(f[0], f[1]); v.y = pack2bf(f[2], f[3]); v.z = pack2bf(f[4], f[5]); v.w = pack2bf(f[6], f[7]);
  return v;
}
__device__ __forceinline__ void unpack4(const uint2& v, float* f) {
  f[0] = bflo(v.x); f[1] = bfhi(v.x); f[2] = bflo(v.y); f[3] = bfhi(v.y);
}
__device__ __forceinline__ uint2 pack4(const float* f) {
  uint2 v;
  v.x = pack2bf(f[0], f[1]); v.y = pack2bf(f[2], f[3]);
  return v;
}

// ---- wave / block reductions --------------------------------------------------------------------
__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  return v;
}
__device__ __forceinline__ float wave_max(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o, 64));
  return v;
}
// block-wide sum for blockDim.x <= 1024 (multiple of 64); `red` is >= 16 floats of LDS. All threads get the result.
__device__ __forceinline__ float block_sum(float v, float* red) {
  v = wave_sum(v);
  const int w = threadIdx.x >> 6, nw = (blockDim.x + 63) >> 6;
  __syncthreads();
  if ((threadIdx.x & 63) == 0) red[w] = v;
  __syncthreads();
  float t = 0.f;
  for (int i = 0; i < nw; ++i) t += red[i];
  return t;
}

// 16-byte global load through a buffer resource: 32-bit per-lane byte offset + wave-uniform scalar byte offset, out-of-range
// offsets read as zero (non-template wrappers: the builtins cannot be instantiated from value-dependent contexts on the host pass)
typedef __attribute__((ext_vector_type(4))) unsigned u32x4_native;
__device__ __forceinline__ uint4 buf_load16(__amdgpu_buffer_rsrc_t rs, unsigned voff, int soff) {
  const u32x4_native v = __builtin_amdgcn_raw_buffer_load_b128(rs, voff, soff, 0);
  return make_uint4(v.x, v.y, v.z, v.w);
}
__device__ __forceinline__ __amdgpu_buffer_rsrc_t make_rsrc(const void* base, unsigned bytes) {
  return __builtin_amdgcn_make_buffer_rsrc((void*)base, 0, (int)bytes, 0x00020000);
}

__device__ __forceinline__ float silu_f(float x) { return x / (1.f + __expf(-x)); }
__device__ __forceinline__ float dsilu_f(float x) {
  const float s = 1.f / (1.f + __expf(-x));
  return s * (1.f + x * (1.f - s));
}
// Exact GELU, x * Phi(x) (F.gelu default, [3P] open_clip ViT / attention.py:428-430 GEGLU).  erf by Abramowitz & Stegun 7.1.26
// (|error| <= 1.5e-7, i.e. fp32 rounding level — libm's erff costs ~40 VALU slots per element, which made the GELU epilogue of the
// ViT's fc1 GEMM 30 % of that kernel: 85 vs 64 us at M4112 N5120 K1280): with z = |x| / sqrt(2), t = 1 / (1 + p z),
// erf(z) = 1 - (a1 t + a2 t^2 + a3 t^3 + a4 t^4 + a5 t^5) exp(-z^2); exp(-z^2) = exp(-x^2 / 2) is the Gaussian the derivative needs anyway.
__device__ __forceinline__ float erf_pos(float z, float e) {       // z >= 0, e = exp(-z*z)
  const float t = __builtin_amdgcn_rcpf(fmaf(0.3275911f, z, 1.f));
  float q = 1.061405429f;
  q = fmaf(q, t, -1.453152027f);
  q = fmaf(q, t, 1.421413741f);
  q = fmaf(q, t, -0.284496736f);
  q = fmaf(q, t, 0.254829592f);
  return fmaf(-q * t, e, 1.f);
}
__device__ __forceinline__ float gelu_f(float x) {
  const float e = __expf(-0.5f * x * x);
  const float er = erf_pos(fabsf(x) * 0.70710678118654752f, e);
  return 0.5f * x * (1.f + copysignf(er, x));
}
__device__ __forceinline__ float dgelu_f(float x) {
  const float e = __expf(-0.5f * x * x);
  const float er = erf_pos(fabsf(x) * 0.70710678118654752f, e);
  const float cdf = 0.5f * (1.f + copysignf(er, x));
  return fmaf(x * 0.39894228040143268f, e, cdf);
}

static inline int cdiv(int a, int b) { return (a + b - 1) / b; }
static inline long long cdivl(long long a, long long b) { return (a + b - 1) / b; }
