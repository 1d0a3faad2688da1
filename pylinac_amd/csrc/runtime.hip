// Error plumbing and version/introspection entry points of the C ABI.
#include <stdarg.h>
#include <stdio.h>

#include "pl_common.h"

static thread_local char g_err[512] = "";

void pl_set_error(const char* fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(g_err, sizeof(g_err), fmt, ap);
  va_end(ap);
}

int pl_check_launch(const char* what) {
  hipError_t e = hipGetLastError();
  if (e != hipSuccess) {
    pl_set_error("%s: HIP launch failed: %s", what, hipGetErrorString(e));
    return PL_ERR_HIP;
  }
  return PL_OK;
}

int pl_cu_count() {
  static std::atomic<int> cached[64];
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) { (void)hipGetLastError(); return 256; }
  int c = cached[dev].load(std::memory_order_relaxed);
  if (c > 0) return c;
  if (hipDeviceGetAttribute(&c, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || c <= 0) {
    (void)hipGetLastError();
    c = 256;
  }
  cached[dev].store(c, std::memory_order_relaxed);
  return c;
}

extern "C" int pl_abi_version(void) { return PL_ABI_VERSION; }

extern "C" const char* pl_last_error(void) { return g_err; }

extern "C" const char* pl_status_string(int status) {
  switch (status) {
    case PL_OK: return "ok";
    case PL_ERR_INVALID_ARG: return "invalid argument";
    case PL_ERR_UNSUPPORTED: return "unsupported dtype/shape";
    case PL_ERR_HIP: return "HIP runtime error";
    default: return "unknown status";
  }
}

extern "C" int pl_device_available(void) {
  int n = 0;
  hipError_t e = hipGetDeviceCount(&n);
  if (e != hipSuccess) {
    (void)hipGetLastError();
    return 0;
  }
  return n > 0 ? 1 : 0;
}
