// runtime.hip -- error reporting + library identification for the C ABI (include/convnet_hip.h).
// Every cn_* entry point returns 0 on success or a negative CN_E* code; the human-readable reason
// is kept in a thread-local buffer that cn_last_error() returns.  Nothing here allocates device
// memory or synchronises the device.
#include "cn_api_internal.h"
#include <stdarg.h>
#include <stdio.h>
#include <string.h>

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
// ... and the names of ALL GEMM-class launches since the log was last cleared, ';'-separated, in launch order: one
// entry point can launch several instantiations (a strided dgrad runs one launch per output-parity class, each
// dispatched on its own reduction length), and measurement code must count launches per kernel the way rocprofv3 does.
static thread_local char g_klog[1024] = "";
static thread_local int g_klog_len = 0;
void cn_set_last_kernel(const char* fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(g_kernel, sizeof(g_kernel), fmt, ap);
  va_end(ap);
  const int n = (int)strlen(g_kernel);
  if (g_klog_len + n + 2 < (int)sizeof(g_klog)) {
    if (g_klog_len > 0) g_klog[g_klog_len++] = ';';
    memcpy(g_klog + g_klog_len, g_kernel, (size_t)n + 1);
    g_klog_len += n;
  }
}
extern "C" const char* cn_last_kernel_name(void) { return g_kernel; }
// clear != 0: empty the log and return ""; clear == 0: the names logged since the last clear
extern "C" const char* cn_kernel_log(int clear) {
  if (clear) { g_klog[0] = 0; g_klog_len = 0; }
  return g_klog;
}

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

// Cross-stream ordering without torch: `to` waits for everything queued on `from` so far.  Events come from a
// ring created once (timing disabled, system-scope fence disabled: both streams are on this device, nothing here has to
// become visible to the host).
// launch plans (plan.hip) log the hand-offs below while one is being recorded
void cn_plan_rec_fork(void* from, void* to);
int cn_plan_rec_wait_mark(int handle, void* to);

#define CN_FORK_EVENTS 256
extern "C" int cn_stream_fork(void* from_, void* to_) {
#ifdef CN_EMULATE
  if (cn_plan_recording) cn_plan_rec_fork(from_, to_);
  return CN_OK;
#else
  static thread_local hipEvent_t ring[CN_FORK_EVENTS];
  static thread_local int made = 0, next = 0;
  if (!made) {
    for (int i = 0; i < CN_FORK_EVENTS; ++i)
      if (hipEventCreateWithFlags(&ring[i], hipEventDisableTiming | hipEventDisableSystemFence) != hipSuccess) { cn_set_error("stream_fork: hipEventCreate failed"); return CN_EHIP; }
    made = 1;
  }
  hipEvent_t e = ring[next];
  next = (next + 1) % CN_FORK_EVENTS;
  if (cn_plan_recording) cn_plan_rec_fork(from_, to_);     // (the event ops below still run: under capture they are the graph's edges)
  if (hipEventRecord(e, (hipStream_t)from_) != hipSuccess || hipStreamWaitEvent((hipStream_t)to_, e, 0) != hipSuccess) {
    cn_set_error("stream_fork: %s", hipGetErrorString(hipGetLastError()));
    return CN_EHIP;
  }
  return CN_OK;
#endif
}

// Marks: arm -> the kernels this thread launches until disarm carry a ring event as their completion event;
// cn_stream_wait_mark makes another stream wait for the last of them.  disarm returns 1 when a kernel took the event.
#ifndef CN_EMULATE
thread_local hipEvent_t cn_tl_stop_event = nullptr;
thread_local int cn_tl_stop_recorded = 0;
thread_local int cn_tl_stop_hold = 0;
thread_local int cn_tl_stop_handle = -1;   // ring handle of the armed event (launch plans record marks by handle)
static thread_local hipEvent_t g_marks[CN_FORK_EVENTS];
static thread_local int g_marks_made = 0, g_mark_next = 0;
#endif
extern "C" int cn_stream_arm(void) {
#ifdef CN_EMULATE
  return 0;
#else
  if (!g_marks_made) {
    for (int i = 0; i < CN_FORK_EVENTS; ++i)
      if (hipEventCreateWithFlags(&g_marks[i], hipEventDisableTiming | hipEventDisableSystemFence) != hipSuccess) {
        cn_set_error("stream_arm: hipEventCreate failed");
        return CN_EHIP;
      }
    g_marks_made = 1;
  }
  const int h = g_mark_next;
  g_mark_next = (g_mark_next + 1) % CN_FORK_EVENTS;
  cn_tl_stop_event = g_marks[h];
  cn_tl_stop_handle = h;
  cn_tl_stop_recorded = 0;
  return h;
#endif
}
extern "C" int cn_stream_disarm(void) {
#ifdef CN_EMULATE
  return 0;
#else
  cn_tl_stop_event = nullptr;
  return cn_tl_stop_recorded;
#endif
}
extern "C" int cn_stream_wait_mark(int handle, void* to_stream) {
#ifdef CN_EMULATE
  (void)handle; (void)to_stream;
  return CN_OK;
#else
  if (!g_marks_made || handle < 0 || handle >= CN_FORK_EVENTS) { cn_set_error("stream_wait_mark: bad handle %d", handle); return CN_EINVAL; }
  if (cn_plan_recording) {
    const int rc = cn_plan_rec_wait_mark(handle, to_stream);
    if (rc != CN_OK) return rc;
  }
  if (hipStreamWaitEvent((hipStream_t)to_stream, g_marks[handle], 0) != hipSuccess) {
    cn_set_error("stream_wait_mark: %s", hipGetErrorString(hipGetLastError()));
    return CN_EHIP;
  }
  return CN_OK;
#endif
}

// Step timer: cn_step_timer_mark(stream, tag) records the next event of a ring behind the stream's work (timing enabled,
// NO system-scope fence: torch's timing events flush the caches at every record - 0.5 % of a 17 ms step when one is
// recorded per step) and remembers the caller's tag with it; cn_step_timer_poll(&ms, &tag_prev, &tag_cur) consumes the
// oldest pair whose events have both completed and returns the time between them with the two marks' tags, without
// waiting (0: nothing complete yet).  Several watchers share the ring: each tags its marks and keeps only the periods
// whose two ends are its own (trainer.EagerWatch); nothing here resets another watcher's marks.
#ifndef CN_EMULATE
#define CN_TIMER_EVENTS 64
static thread_local hipEvent_t g_timer[CN_TIMER_EVENTS];
static thread_local long long g_timer_tag[CN_TIMER_EVENTS];
static thread_local int g_timer_made = 0;
static thread_local long long g_timer_head = 0, g_timer_tail = 0;   // [head, tail): recorded, not yet consumed
#endif
extern "C" int cn_step_timer_mark(void* stream, long long tag) {
#ifdef CN_EMULATE
  (void)stream; (void)tag;
  return CN_OK;
#else
  if (!g_timer_made) {
    for (int i = 0; i < CN_TIMER_EVENTS; ++i)
      if (hipEventCreateWithFlags(&g_timer[i], hipEventDisableSystemFence) != hipSuccess) { cn_set_error("step_timer: hipEventCreate failed"); return CN_EHIP; }
    g_timer_made = 1;
  }
  if (g_timer_tail - g_timer_head >= CN_TIMER_EVENTS) g_timer_head = g_timer_tail - 1;   // never polled: keep the newest
  if (hipEventRecord(g_timer[g_timer_tail % CN_TIMER_EVENTS], (hipStream_t)stream) != hipSuccess) {
    cn_set_error("step_timer: %s", hipGetErrorString(hipGetLastError()));
    return CN_EHIP;
  }
  g_timer_tag[g_timer_tail % CN_TIMER_EVENTS] = tag;
  ++g_timer_tail;
  return CN_OK;
#endif
}
extern "C" int cn_step_timer_poll(float* period_ms, long long* tag_prev, long long* tag_cur) {
#ifdef CN_EMULATE
  (void)period_ms; (void)tag_prev; (void)tag_cur;
  return 0;
#else
  if (period_ms == nullptr || !g_timer_made || g_timer_tail - g_timer_head < 2) return 0;
  const int ia = (int)(g_timer_head % CN_TIMER_EVENTS), ib = (int)((g_timer_head + 1) % CN_TIMER_EVENTS);
  if (hipEventQuery(g_timer[ib]) != hipSuccess) { (void)hipGetLastError(); return 0; }   // hipErrorNotReady is not an error here
  if (tag_prev) *tag_prev = g_timer_tag[ia];
  if (tag_cur) *tag_cur = g_timer_tag[ib];
  ++g_timer_head;
  if (hipEventElapsedTime(period_ms, g_timer[ia], g_timer[ib]) != hipSuccess) {
    (void)hipGetLastError();
    *period_ms = -1.0f;      // the pair is consumed all the same: the caller sees its tags and a negative period
  }
  return 1;
#endif
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
#ifndef CN_SRC_HASH
#define CN_SRC_HASH "unknown"
#endif
  return "convnet_hip gfx950 (CDNA4 / MI355X) HIP build; src " CN_SRC_HASH;
#endif
}
