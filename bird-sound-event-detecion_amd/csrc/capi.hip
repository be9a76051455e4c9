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
extern "C" int bsed_abi_version(void) { return 2; }

// ----------------------------------------------------------------------------------------------
// Device-resident step state (HIP-graph replays of a train step, engine.SEDTrainer.capture): a captured launch bakes
// its scalar arguments, so what changes from step to step -- the dropout seed and the optimizer's step count -- is ALSO
// read from device memory: every kernel that takes a seed adds *seed_add to it, the Adam kernel adds *step_add to its
// step, and bsed_step_state_advance (a node of the graph) bumps both.  Eager steps leave the pointers null (or the
// values zero): same effective seeds and steps, same bits.
// ----------------------------------------------------------------------------------------------
static const uint64_t* g_seed_add = nullptr;
static const int* g_step_add = nullptr;
const uint64_t* bsed_seed_add_ptr() { return g_seed_add; }
const int* bsed_step_add_ptr() { return g_step_add; }
extern "C" int bsed_set_step_state(const void* seed_add_dev, const void* step_add_dev) {
  g_seed_add = (const uint64_t*)seed_add_dev;
  g_step_add = (const int*)step_add_dev;
  return BSED_OK;
}
__global__ void step_state_advance_kernel(uint64_t* seed_add, int* step_add, uint64_t seed_inc, int step_inc) {
  if (threadIdx.x == 0 && blockIdx.x == 0) { *seed_add += seed_inc; *step_add += step_inc; }
}
extern "C" int bsed_step_state_advance(void* seed_add_dev, void* step_add_dev, uint64_t seed_inc, int step_inc, void* stream) {
  BSED_CHECK_ARG(seed_add_dev && step_add_dev, "bsed_step_state_advance: null state");
  hipLaunchKernelGGL(step_state_advance_kernel, dim3(1), dim3(64), 0, (hipStream_t)stream, (uint64_t*)seed_add_dev,
                     (int*)step_add_dev, seed_inc, step_inc);
  BSED_LAUNCH_CHECK();
  return BSED_OK;
}
