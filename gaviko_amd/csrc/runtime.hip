// Error plumbing and device checks shared by every C-ABI entry point.
#include "common.hpp"
#include "../../include/gaviko_hip.h"
#include <cstdarg>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <memory>
#include <mutex>
#include <vector>

namespace gvk {
static thread_local char g_err[512] = "";

int set_error(int code, const char* fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(g_err, sizeof(g_err), fmt, ap);
  va_end(ap);
  return code;
}

int check_launch(const char* what) {
  hipError_t e = hipGetLastError();
  if (e != hipSuccess) return set_error(-1, "%s: launch failed: %s", what, hipGetErrorString(e));
  return 0;
}

// ---- launch plans ---------------------------------------------------------------------------------------------------------
struct Plan {
  std::vector<std::function<void()>> nodes;
  std::vector<hipEvent_t> events;
  ~Plan() {
    for (hipEvent_t e : events) (void)hipEventDestroy(e);
  }
};
static thread_local Plan* g_rec = nullptr;
static std::mutex g_plans_mu;
static std::vector<std::unique_ptr<Plan>> g_plans;

static bool g_plan_timing = false;      // gvk_plan_set_timing: events of plans recorded from now on carry timestamps
bool plan_recording() { return g_rec != nullptr; }
void plan_push(std::function<void()>&& node) { g_rec->nodes.push_back(std::move(node)); }

#ifdef GVK_DIAG
__global__ void nop_kernel() {}
// Diagnostics, diag library only (include/gaviko_hip_diag.h; GAVIKO_HIP_ABLATE=sidenop|locnop|gpanop): every launch on a stream the
// engine registered with gvk_plan_nop_stream keeps its place, stream and events but runs an empty kernel -- separates what the side
// streams cost the main one in dispatch and synchronisation from what they cost in CUs and bandwidth.  Results are garbage.
static std::vector<hipStream_t> g_nop_streams;
void plan_push_launch(hipStream_t stream, std::function<void()>&& node) {
  for (hipStream_t s : g_nop_streams)
    if (s == stream) {
      g_rec->nodes.push_back([=]() { hipLaunchKernelGGL(nop_kernel, dim3(1), dim3(64), 0, stream); });
      return;
    }
  g_rec->nodes.push_back(std::move(node));
}
#else
void plan_push_launch(hipStream_t, std::function<void()>&& node) { g_rec->nodes.push_back(std::move(node)); }
#endif

__global__ void seed_advance_kernel(unsigned long long* seed, unsigned long long inc) { seed[0] += inc; }
// zero / copy as ordinary kernels: hipMemsetAsync issued inside a torch stream capture was executed immediately instead of
// becoming a graph node on this runtime (the replayed hipGraph then ran on stale buffers), and kernels are recorded into a
// launch plan like everything else.
__global__ __launch_bounds__(256) void fill_u32_kernel(unsigned* p, unsigned v, long n4, long n) {
  const long stride = (long)gridDim.x * 256;
  for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n4; i += stride) ((uint4*)p)[i] = uint4{v, v, v, v};
  for (long i = 4 * n4 + (long)blockIdx.x * 256 + threadIdx.x; i < n; i += stride) p[i] = v;
}
__global__ __launch_bounds__(256) void copy_u32_kernel(unsigned* dst, const unsigned* src, long n4, long n) {
  const long stride = (long)gridDim.x * 256;
  for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n4; i += stride) ((uint4*)dst)[i] = ((const uint4*)src)[i];
  for (long i = 4 * n4 + (long)blockIdx.x * 256 + threadIdx.x; i < n; i += stride) dst[i] = src[i];
}
__global__ void scale_kernel(float* x, float alpha, long n) {
  const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) x[i] *= alpha;
}
}  // namespace gvk

#ifdef GVK_DIAG
extern "C" int gvk_plan_nop_stream(void* stream) {
  gvk::g_nop_streams.push_back((hipStream_t)stream);
  return 0;
}
extern "C" int gvk_plan_nop_clear(void) {
  gvk::g_nop_streams.clear();
  return 0;
}
#endif

extern "C" int gvk_plan_begin(void) {
  using namespace gvk;
  GVK_REQUIRE(g_rec == nullptr, "gvk_plan_begin: a plan is already being recorded on this thread");
  g_rec = new Plan();
  return 0;
}

extern "C" int gvk_plan_abort(void) {
  delete gvk::g_rec;
  gvk::g_rec = nullptr;
  return 0;
}

extern "C" int gvk_plan_end(void) {
  using namespace gvk;
  GVK_REQUIRE(g_rec != nullptr, "gvk_plan_end: no plan is being recorded");
  std::lock_guard<std::mutex> lk(g_plans_mu);
  g_plans.emplace_back(g_rec);
  g_rec = nullptr;
  return (int)g_plans.size() - 1;
}

extern "C" int gvk_plan_size(int plan) {
  using namespace gvk;
  std::lock_guard<std::mutex> lk(g_plans_mu);
  GVK_REQUIRE(plan >= 0 && plan < (int)g_plans.size() && g_plans[plan], "gvk_plan_size: no such plan %d", plan);
  return (int)g_plans[plan]->nodes.size();
}

extern "C" int gvk_plan_replay(int plan) {
  using namespace gvk;
  Plan* p = nullptr;
  {
    std::lock_guard<std::mutex> lk(g_plans_mu);
    GVK_REQUIRE(plan >= 0 && plan < (int)g_plans.size() && g_plans[plan], "gvk_plan_replay: no such plan %d", plan);
    p = g_plans[plan].get();
  }
  GVK_REQUIRE(g_rec == nullptr, "gvk_plan_replay: cannot replay while recording");
  for (auto& node : p->nodes) node();
  return check_launch("plan_replay");
}

extern "C" int gvk_plan_free(int plan) {
  using namespace gvk;
  std::lock_guard<std::mutex> lk(g_plans_mu);
  GVK_REQUIRE(plan >= 0 && plan < (int)g_plans.size() && g_plans[plan], "gvk_plan_free: no such plan %d", plan);
  g_plans[plan].reset();
  return 0;
}

extern "C" int gvk_plan_set_timing(int on) {
  gvk::g_plan_timing = on != 0;
  return 0;
}

extern "C" int gvk_plan_event_elapsed(int plan, int e0, int e1, float* ms) {
  using namespace gvk;
  std::lock_guard<std::mutex> lk(g_plans_mu);
  GVK_REQUIRE(plan >= 0 && plan < (int)g_plans.size() && g_plans[plan], "gvk_plan_event_elapsed: no such plan %d", plan);
  Plan* p = g_plans[plan].get();
  GVK_REQUIRE(e0 >= 0 && e1 >= 0 && e0 < (int)p->events.size() && e1 < (int)p->events.size() && ms, "gvk_plan_event_elapsed: bad event ids");
  hipError_t e = hipEventElapsedTime(ms, p->events[e0], p->events[e1]);
  if (e != hipSuccess) return set_error(-1, "hipEventElapsedTime: %s (timing needs GAVIKO_HIP_PLAN_TIMING=1 at record time)", hipGetErrorString(e));
  return 0;
}

static int plan_event_record_impl(void* stream, bool force_sys_fence);
extern "C" int gvk_plan_event_record(void* stream) { return plan_event_record_impl(stream, false); }
// The same with the system-scope fence kept: for events that another DEVICE's reads are ordered behind (the gradient buckets an
// all-reduce sends to peer GPUs over xGMI).
extern "C" int gvk_plan_event_record_fenced(void* stream) { return plan_event_record_impl(stream, true); }

static int plan_event_record_impl(void* stream, bool force_sys_fence) {
  using namespace gvk;
  GVK_REQUIRE(g_rec != nullptr, "gvk_plan_event_record: only valid while a plan is being recorded");
  static const bool env_timing = getenv("GAVIKO_HIP_PLAN_TIMING") != nullptr;  // diagnostics: tools/plan_marks.py
  hipEvent_t ev;
  // Plan events only order streams of ONE device: kernel boundaries already carry the device-scope release/acquire, so the
  // system-scope fence (an L2 writeback + invalidate per record, and refetches for whatever runs next) is switched off.
  // GAVIKO_HIP_EVENT_FENCE=1 restores the default events.
  static const bool sys_fence = diag_env("GAVIKO_HIP_EVENT_FENCE") != nullptr;
  const unsigned flags = ((env_timing || g_plan_timing) ? hipEventDefault : hipEventDisableTiming) | ((sys_fence || force_sys_fence) ? 0u : hipEventDisableSystemFence);
  hipError_t e = hipEventCreateWithFlags(&ev, flags);
  if (e != hipSuccess) return set_error(-1, "hipEventCreate: %s", hipGetErrorString(e));
  g_rec->events.push_back(ev);
  hipStream_t s = (hipStream_t)stream;
  g_rec->nodes.push_back([=]() { (void)hipEventRecord(ev, s); });
  e = hipEventRecord(ev, s);
  if (e != hipSuccess) return set_error(-1, "hipEventRecord: %s", hipGetErrorString(e));
  return (int)g_rec->events.size() - 1;
}

// Make `stream` (any stream, e.g. the collective's) wait for event `event` of a recorded plan's most recent replay -- issued now, not
// recorded.  This is how a gradient bucket's all-reduce is ordered behind the kernels that finalise it without cutting the plan.
extern "C" int gvk_plan_event_stream_wait(int plan, int event, void* stream) {
  using namespace gvk;
  hipEvent_t ev;
  {
    std::lock_guard<std::mutex> lk(g_plans_mu);
    GVK_REQUIRE(plan >= 0 && plan < (int)g_plans.size() && g_plans[plan], "gvk_plan_event_stream_wait: no such plan %d", plan);
    Plan* p = g_plans[plan].get();
    GVK_REQUIRE(event >= 0 && event < (int)p->events.size(), "gvk_plan_event_stream_wait: no such event %d", event);
    ev = p->events[event];
  }
  hipError_t e = hipStreamWaitEvent((hipStream_t)stream, ev, 0);
  if (e != hipSuccess) return set_error(-1, "hipStreamWaitEvent: %s", hipGetErrorString(e));
  return 0;
}

extern "C" int gvk_plan_event_wait(void* stream, int event) {
  using namespace gvk;
  GVK_REQUIRE(g_rec != nullptr, "gvk_plan_event_wait: only valid while a plan is being recorded");
  GVK_REQUIRE(event >= 0 && event < (int)g_rec->events.size(), "gvk_plan_event_wait: no such event %d", event);
  hipEvent_t ev = g_rec->events[event];
  hipStream_t s = (hipStream_t)stream;
  g_rec->nodes.push_back([=]() { (void)hipStreamWaitEvent(s, ev, 0); });
  hipError_t e = hipStreamWaitEvent(s, ev, 0);
  if (e != hipSuccess) return set_error(-1, "hipStreamWaitEvent: %s", hipGetErrorString(e));
  return 0;
}

extern "C" int gvk_memset_async(void* ptr, int value, size_t bytes, void* stream) {
  using namespace gvk;
  GVK_REQUIRE(ptr != nullptr || bytes == 0, "gvk_memset_async: null pointer");
  if (bytes == 0) return 0;
  GVK_REQUIRE(((uintptr_t)ptr & 15) == 0 && bytes % 4 == 0, "gvk_memset_async: pointer must be 16-byte aligned and the size a multiple of 4");
  const unsigned b = (unsigned)value & 0xFFu, v = b | (b << 8) | (b << 16) | (b << 24);
  const long n = (long)(bytes / 4), n4 = n / 4;
  long blocks = (n4 + 255) / 256;
  blocks = blocks < 1 ? 1 : (blocks > 2048 ? 2048 : blocks);
  GVK_LAUNCH(fill_u32_kernel, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, (unsigned*)ptr, v, n4, n);
  return check_launch("memset_async");
}

extern "C" int gvk_seed_advance(void* seed, uint64_t inc, void* stream) {
  using namespace gvk;
  GVK_REQUIRE(seed != nullptr, "gvk_seed_advance: null pointer");
  GVK_LAUNCH(seed_advance_kernel, dim3(1), dim3(1), 0, (hipStream_t)stream, (unsigned long long*)seed, (unsigned long long)inc);
  return check_launch("seed_advance");
}

extern "C" int gvk_scale_f32(float* x, float alpha, long n, void* stream) {
  using namespace gvk;
  GVK_REQUIRE(x != nullptr && n >= 0, "gvk_scale_f32: bad arguments");
  if (n == 0) return 0;
  GVK_LAUNCH(scale_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, (hipStream_t)stream, x, alpha, n);
  return check_launch("scale_f32");
}

extern "C" const char* gvk_last_error(void) { return gvk::g_err; }
extern "C" int gvk_abi_version(void) { return 12; }

extern "C" int gvk_device_check(void) {
  int dev = 0;
  hipError_t e = hipGetDevice(&dev);
  if (e != hipSuccess) return gvk::set_error(-1, "hipGetDevice: %s", hipGetErrorString(e));
  hipDeviceProp_t prop;
  e = hipGetDeviceProperties(&prop, dev);
  if (e != hipSuccess) return gvk::set_error(-1, "hipGetDeviceProperties: %s", hipGetErrorString(e));
  if (strncmp(prop.gcnArchName, "gfx950", 6) != 0)
    return gvk::set_error(-4, "libgaviko_hip is built for gfx950 only; device %d is %s", dev, prop.gcnArchName);
  return 950;
}
