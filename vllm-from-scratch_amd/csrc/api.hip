// Error reporting and version entry points of libmi355_paged.so.
#include <stdarg.h>
#include <stdio.h>

#include "common.h"

namespace mi355 {

static thread_local char g_err[512] = "";

int fail(int code, const char* fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(g_err, sizeof(g_err), fmt, ap);
  va_end(ap);
  return code;
}

int check_launch(const char* what) {
  hipError_t e = hipGetLastError();
  if (e == hipSuccess) return 0;
  return fail((int)e, "%s: %s", what, hipGetErrorString(e));
}

}  // namespace mi355

extern "C" int mi355_abi_version(void) { return MI355_ABI_VERSION; }

extern "C" const char* mi355_last_error(void) { return mi355::g_err; }
