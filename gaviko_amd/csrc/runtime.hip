// Error plumbing and device checks shared by every C-ABI entry point.
#include "common.hpp"
#include "../../include/gaviko_hip.h"
#include <cstdarg>
#include <cstdio>
#include <cstring>

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
}  // namespace gvk

extern "C" const char* gvk_last_error(void) { return gvk::g_err; }
extern "C" int gvk_abi_version(void) { return 1; }

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
