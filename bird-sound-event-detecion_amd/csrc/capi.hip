// Error plumbing + build info of libbsed.so.
#include "bsed_common.h"
#include "../../include/bsed.h"
#include <stdarg.h>

static thread_local char g_err[512] = "";

void bsed_set_error(const char* fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(g_err, sizeof(g_err), fmt, ap);
  va_end(ap);
}

extern "C" const char* bsed_last_error(void) { return g_err; }
extern "C" const char* bsed_build_info(void) { return "libbsed gfx950 fp32 (hand-written HIP, MFMA f32)"; }
extern "C" int bsed_abi_version(void) { return 1; }
