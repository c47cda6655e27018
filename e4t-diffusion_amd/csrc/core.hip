// Library plumbing: version, last-error string, device info, MFMA layout probe.
#include "common.h"
#include "../../include/e4t_hip.h"

static thread_local char g_err[512] = "";

extern "C" void e4t_set_error(const char* msg) {
  strncpy(g_err, msg, sizeof(g_err) - 1);
  g_err[sizeof(g_err) - 1] = 0;
}
extern "C" const char* e4t_last_error(void) { return g_err; }
extern "C" int e4t_version(void) { return 100; }
extern "C" int e4t_build_flags(void) {
#ifdef E4T_EXPERIMENTAL
  return E4T_BUILD_EXPERIMENTAL;
#else
  return 0;
#endif
}

// ---- launch log ------------------------------------------------------------------------------------
#include <stdarg.h>
#include <stdlib.h>
static FILE* g_log = nullptr;
static int g_log_state = -1;      // -1: environment not looked at yet
extern "C" int e4t_set_launch_log(const char* path) {
  if (g_log) { fclose(g_log); g_log = nullptr; }
  g_log_state = 0;
  if (path && path[0]) {
    g_log = fopen(path, "w");
    if (!g_log) E4T_FAIL(-2, "set_launch_log: cannot open %s", path);
    g_log_state = 1;
  }
  return 0;
}
extern "C" int e4t_launch_log_enabled(void) {
  if (g_log_state < 0) {
    const char* p = getenv("E4T_LAUNCH_LOG");
    if (p && p[0]) e4t_set_launch_log(p); else g_log_state = 0;
  }
  return g_log_state;
}
extern "C" void e4t_launch_logf(const char* fmt, ...) {
  if (!g_log) return;
  va_list ap;
  va_start(ap, fmt);
  vfprintf(g_log, fmt, ap);
  va_end(ap);
  fputc('\n', g_log);
  fflush(g_log);
}

extern "C" int e4t_device_info(char* arch, int arch_len, int* cu_count) {
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess) E4T_FAIL(-19, "device_info: no HIP device");
  hipDeviceProp_t prop;
  if (hipGetDeviceProperties(&prop, dev) != hipSuccess) E4T_FAIL(-19, "device_info: hipGetDeviceProperties failed");
  if (arch && arch_len > 0) {
    strncpy(arch, prop.gcnArchName, arch_len - 1);
    arch[arch_len - 1] = 0;
  }
  if (cu_count) *cu_count = prop.multiProcessorCount;
  return 0;
}

// Probe of v_mfma_f32_32x32x16_bf16 operand/accumulator placement (one wave).
//   pass 0: A_lane(l)[0] = l+1 for l < 32, B_lane(l)[0] = 1 for l < 32  -> D[i][n] = i+1  (row map)
//   pass 1: A_lane(l)[0] = 1,              B_lane(l)[0] = l+1           -> D[i][n] = n+1  (col map)
// Expected: out_rows[l][r] = (r&3) + 8*(r>>2) + 4*(l>>5) + 1 ; out_cols[l][r] = (l&31) + 1.
__global__ void probe_kernel(float* out_rows, float* out_cols) {
  const int l = threadIdx.x;
  for (int pass = 0; pass < 2; ++pass) {
    bf16x8 a, b;
    for (int j = 0; j < 8; ++j) { a[j] = (__bf16)0.f; b[j] = (__bf16)0.f; }
    if (l < 32) {
      a[0] = (__bf16)(pass == 0 ? (float)(l + 1) : 1.f);
      b[0] = (__bf16)(pass == 0 ? 1.f : (float)(l + 1));
    }
    f32x16 c;
    for (int r = 0; r < 16; ++r) c[r] = 0.f;
    c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c, 0, 0, 0);
    float* o = pass == 0 ? out_rows : out_cols;
    for (int r = 0; r < 16; ++r) o[l * 16 + r] = c[r];
  }
}

extern "C" int e4t_probe_mfma_layout(float* out_rows, float* out_cols, e4t_stream stream) {
  E4T_REQUIRE(out_rows && out_cols, "probe: null output");
  hipLaunchKernelGGL(probe_kernel, dim3(1), dim3(64), 0, (hipStream_t)stream, out_rows, out_cols);
  E4T_CHECK_LAUNCH("probe_kernel");
  return 0;
}
