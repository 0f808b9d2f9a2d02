// Error state and version of libvlhip.so (see include/vlhip.h for the conventions).
#include <stdarg.h>
#include <stdio.h>

#include "common.h"
#include "../../include/vlhip.h"

namespace {
thread_local char g_err[512] = "";
}

int vl_set_error(int code, const char* fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(g_err, sizeof(g_err), fmt, ap);
  va_end(ap);
  return code;
}

extern "C" const char* vl_last_error(void) { return g_err; }
extern "C" int vl_version(void) { return 100; /* 0.1.0 */ }

// zero-fill on the caller's stream (hipMemsetAsync): padding of scratch buffers without touching another stream
extern "C" int vl_memset_zero(void* p, int64_t bytes, void* stream) {
  VL_CHECK_ARG(p && bytes >= 0, "vl_memset_zero: bad arguments");
  hipError_t e = hipMemsetAsync(p, 0, (size_t)bytes, (hipStream_t)stream);
  if (e != hipSuccess) return vl_set_error(-3, "vl_memset_zero: %s", hipGetErrorString(e));
  return 0;
}
