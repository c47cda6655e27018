// Data path: the reference's per-sample CPU transform
//   SmallestMaxSize(size, INTER_AREA) -> RandomCrop(size) -> HorizontalFlip -> x/127.5 - 1 -> CHW fp32
// (pretrain_e4t.py:137-144,174-177) as ONE kernel over a batch of raw uint8 RGB images that were copied to
// the device as they came out of the decoder.  Only the cropped window of the resized image is ever
// computed: one thread per output pixel walks its source cell (about (H/newH)*(W/newW) pixels).
//
// Byte-exact arithmetic: the resize reproduces OpenCV's 8-bit INTER_AREA result, which depends on the order
// of float operations — so this file is compiled with floating-point contraction OFF (no FMA) and the
// accumulation order is the one of resize.cpp (x pass in table order, then y pass).  HBM-bound and tiny:
// ~scale^2*3 B read + 12 B written per output pixel.
#include "common.h"
#include "../../include/e4t_hip.h"

#pragma clang fp contract(off)

namespace {

struct AreaCell {          // computeResizeAreaTab for one destination index
  int s_first, s_lo, s_hi, s_last;     // optional left partial | full cells [s_lo, s_hi) | optional right partial
  float a_first, a_mid, a_last;
  bool has_first, has_last;
};

__device__ inline AreaCell area_cell(int ssize, double scale, int d) {
  AreaCell c;
  const double f1 = d * scale;
  const double f2 = f1 + scale;
  const double cell = fmin(scale, (double)ssize - f1);
  int s1 = (int)ceil(f1), s2 = (int)floor(f2);
  s2 = min(s2, ssize - 1);
  s1 = min(s1, s2);
  c.has_first = (s1 - f1) > 1e-3;
  c.s_first = s1 - 1;
  c.a_first = (float)((s1 - f1) / cell);
  c.s_lo = s1;
  c.s_hi = s2;
  c.a_mid = (float)(1.0 / cell);
  c.has_last = (f2 - s2) > 1e-3;
  c.s_last = s2;
  c.a_last = (float)(fmin(fmin(f2 - s2, 1.0), cell) / cell);
  return c;
}

struct F3 { float r, g, b; };

__device__ inline void acc(F3& v, const unsigned char* p, float a) {
  v.r = v.r + (float)p[0] * a;
  v.g = v.g + (float)p[1] * a;
  v.b = v.b + (float)p[2] * a;
}

__device__ inline F3 row_pass(const unsigned char* row, const AreaCell& cx) {
  F3 v = {0.f, 0.f, 0.f};
  if (cx.has_first) acc(v, row + 3 * cx.s_first, cx.a_first);
  for (int s = cx.s_lo; s < cx.s_hi; ++s) acc(v, row + 3 * s, cx.a_mid);
  if (cx.has_last) acc(v, row + 3 * cx.s_last, cx.a_last);
  return v;
}

__device__ inline void yacc(F3& s, const F3& b, float beta) {
  s.r = s.r + beta * b.r;
  s.g = s.g + beta * b.g;
  s.b = s.b + beta * b.b;
}

__device__ inline int sat_u8(float v) {           // saturate_cast<uchar>(float): cvRound (half to even), clamp
  int i = __float2int_rn(v);
  return min(max(i, 0), 255);
}

// bilinear coefficients of the "area mode" used when enlarging (fixed point, 11 bits)
__device__ inline void lin_coef(int ssize, double scale, double inv, int d, int& s, int& c0, int& c1) {
  s = (int)floor(d * scale);
  float f = (float)((d + 1) - (s + 1) * inv);
  f = f <= 0.f ? 0.f : f - floorf(f);
  if (s < 0) { f = 0.f; s = 0; }
  if (s >= ssize - 1) { f = 0.f; s = ssize - 1; }
  const float a0 = 1.f - f;
  c0 = min(max(__float2int_rn(a0 * 2048.f), -32768), 32767);
  c1 = min(max(__float2int_rn(f * 2048.f), -32768), 32767);
}

__global__ __launch_bounds__(256) void image_prep_kernel(const unsigned char* __restrict__ pool, const long long* __restrict__ table,
                                                          float* __restrict__ out, int S) {
  const int x = blockIdx.x * 64 + threadIdx.x;
  const int y = blockIdx.y * 4 + threadIdx.y;
  const int b = blockIdx.z;
  if (x >= S || y >= S) return;
  const long long* t = table + (long long)b * 8;
  const unsigned char* img = pool + t[0];
  const int H = (int)t[1], W = (int)t[2], nH = (int)t[3], nW = (int)t[4];
  const int ry = (int)t[5] + y;
  const int rx = (int)t[6] + (t[7] ? S - 1 - x : x);
  const long long pitch = 3ll * W;
  int u0, u1, u2;

  if (nH == H && nW == W) {                                   // SmallestMaxSize leaves the image untouched (scale == 1)
    const unsigned char* p = img + ry * pitch + 3 * rx;
    u0 = p[0]; u1 = p[1]; u2 = p[2];
  } else {
    const double sx = (double)W / nW, sy = (double)H / nH;
    if (sx >= 1.0 && sy >= 1.0) {
      const int ix = (int)rint(sx), iy = (int)rint(sy);
      if (fabs(sx - ix) < 2.220446049250313e-16 && fabs(sy - iy) < 2.220446049250313e-16) {
        int a0 = 0, a1 = 0, a2 = 0;                          // integer factors: plain box sums
        for (int j = 0; j < iy; ++j) {
          const unsigned char* p = img + (long long)(ry * iy + j) * pitch + 3ll * rx * ix;
          for (int i = 0; i < ix; ++i) { a0 += p[3 * i]; a1 += p[3 * i + 1]; a2 += p[3 * i + 2]; }
        }
        if (ix == 2 && iy == 2) {
          u0 = (a0 + 2) >> 2; u1 = (a1 + 2) >> 2; u2 = (a2 + 2) >> 2;
        } else {
          const float sc = 1.f / (float)(ix * iy);
          u0 = sat_u8((float)a0 * sc); u1 = sat_u8((float)a1 * sc); u2 = sat_u8((float)a2 * sc);
        }
      } else {
        const AreaCell cx = area_cell(W, sx, rx);
        const AreaCell cy = area_cell(H, sy, ry);
        F3 s = {0.f, 0.f, 0.f};
        if (cy.has_first) yacc(s, row_pass(img + cy.s_first * pitch, cx), cy.a_first);
        for (int r = cy.s_lo; r < cy.s_hi; ++r) yacc(s, row_pass(img + r * pitch, cx), cy.a_mid);
        if (cy.has_last) yacc(s, row_pass(img + cy.s_last * pitch, cx), cy.a_last);
        u0 = sat_u8(s.r); u1 = sat_u8(s.g); u2 = sat_u8(s.b);
      }
    } else {
      int xs, xa0, xa1, ys, yb0, yb1;
      lin_coef(W, sx, 1.0 / sx, rx, xs, xa0, xa1);
      lin_coef(H, sy, 1.0 / sy, ry, ys, yb0, yb1);
      const int xs1 = min(xs + 1, W - 1), ys1 = min(ys + 1, H - 1);
      const unsigned char* p0 = img + ys * pitch;
      const unsigned char* p1 = img + ys1 * pitch;
      int u[3];
      for (int c = 0; c < 3; ++c) {
        const int r0 = p0[3 * xs + c] * xa0 + p0[3 * xs1 + c] * xa1;
        const int r1 = p1[3 * xs + c] * xa0 + p1[3 * xs1 + c] * xa1;
        const int v = (((yb0 * (r0 >> 4)) >> 16) + ((yb1 * (r1 >> 4)) >> 16) + 2) >> 2;
        u[c] = min(max(v, 0), 255);
      }
      u0 = u[0]; u1 = u[1]; u2 = u[2];
    }
  }
  const long long plane = (long long)S * S;
  float* o = out + (long long)b * 3 * plane + (long long)y * S + x;
  o[0] = (float)((double)u0 / 127.5 - 1.0);
  o[plane] = (float)((double)u1 / 127.5 - 1.0);
  o[2 * plane] = (float)((double)u2 / 127.5 - 1.0);
}

}  // namespace

extern "C" int e4t_image_prep(const void* pool, const long long* table, float* out, int B, int S, e4t_stream stream) {
  E4T_REQUIRE(pool && table && out, "image_prep: null pointer");
  E4T_REQUIRE(B > 0 && S > 0 && B <= 65535, "image_prep: bad batch / size");
  dim3 grid((S + 63) / 64, (S + 3) / 4, B), block(64, 4);
  hipLaunchKernelGGL(image_prep_kernel, grid, block, 0, (hipStream_t)stream, (const unsigned char*)pool, table, out, S);
  E4T_CHECK_LAUNCH("image_prep_kernel");
  return 0;
}
