// runtime.hip -- error reporting + library identification for the C ABI (include/convnet_hip.h).
// Every cn_* entry point returns 0 on success or a negative CN_E* code; the human-readable reason
// is kept in a thread-local buffer that cn_last_error() returns.  Nothing here allocates device
// memory or synchronises the device.
#include "cn_api_internal.h"
#include <stdarg.h>
#include <stdio.h>

static thread_local char g_err[512] = "";

void cn_set_error(const char* fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(g_err, sizeof(g_err), fmt, ap);
  va_end(ap);
}

int cn_check_launch(const char* what) {
#ifdef CN_EMULATE
  (void)what;
  return CN_OK;
#else
  hipError_t e = hipGetLastError();
  if (e != hipSuccess) {
    cn_set_error("%s: HIP launch failed: %s", what, hipGetErrorString(e));
    return CN_EHIP;
  }
  return CN_OK;
#endif
}

extern "C" const char* cn_last_error(void) { return g_err; }

// Name (as rocprofv3 prints it, without "void " and the parameter list) of the GEMM-class kernel the most recent
// cn_conv2d_* / cn_conv2d_wgrad call of this thread launched: the dispatchers choose an instantiation per
// shape, and measurement code labels its timings with this instead of mirroring the heuristics.
static thread_local char g_kernel[160] = "";
void cn_set_last_kernel(const char* fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(g_kernel, sizeof(g_kernel), fmt, ap);
  va_end(ap);
}
extern "C" const char* cn_last_kernel_name(void) { return g_kernel; }

// Tuning knobs (kernel variant selection for A/B measurements; never change results).
#include <string.h>
#define CN_MAX_OPTS 16
static char g_opt_name[CN_MAX_OPTS][32];
static int g_opt_val[CN_MAX_OPTS];
static int g_nopts = 0;
extern "C" int cn_set_option(const char* name, int value) {
  for (int i = 0; i < g_nopts; ++i)
    if (strcmp(g_opt_name[i], name) == 0) { g_opt_val[i] = value; return CN_OK; }
  if (g_nopts >= CN_MAX_OPTS || strlen(name) >= 32) { cn_set_error("set_option: table full / name too long"); return CN_EINVAL; }
  strcpy(g_opt_name[g_nopts], name);
  g_opt_val[g_nopts++] = value;
  return CN_OK;
}
int cn_get_option(const char* name, int dflt) {
  for (int i = 0; i < g_nopts; ++i)
    if (strcmp(g_opt_name[i], name) == 0) return g_opt_val[i];
  return dflt;
}

extern "C" int cn_is_emulator(void) {
#ifdef CN_EMULATE
  return 1;
#else
  return 0;
#endif
}

extern "C" const char* cn_build_info(void) {
#ifdef CN_EMULATE
  return "convnet_hip TEST-ONLY SIMT emulator build (host C++)";
#else
  return "convnet_hip gfx950 (CDNA4 / MI355X) HIP build";
#endif
}
