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
extern "C" const char* bsed_build_info(void) { return "libbsed gfx950: fp32 storage/accumulation, split-fp32 (bf16x3) contractions on v_mfma_f32_32x32x16_bf16 by default, "
         "exact-fp32 v_mfma_f32_32x32x2_f32 kernels selectable (hand-written HIP)"; }
extern "C" int bsed_abi_version(void) { return 1; }
