// Shared device/host helpers for the bsed HIP library (gfx950 / CDNA4 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>

#define BSED_OK 0
#define BSED_ERR_ARG (-1)
#define BSED_ERR_HIP (-2)
#define BSED_ERR_STATE (-3)

void bsed_set_error(const char* fmt, ...);
// device-resident step state (capi.hip): null in eager mode
const uint64_t* bsed_seed_add_ptr();
const int* bsed_step_add_ptr();

#define BSED_CHECK_ARG(cond, ...)                 \
  do {                                            \
    if (!(cond)) {                                \
      bsed_set_error(__VA_ARGS__);                \
      return BSED_ERR_ARG;                        \
    }                                             \
  } while (0)

#define BSED_LAUNCH_CHECK()                                                         \
  do {                                                                              \
    hipError_t e__ = hipGetLastError();                                             \
    if (e__ != hipSuccess) {                                                        \
      bsed_set_error("%s:%d HIP launch error: %s", __FILE__, __LINE__,              \
                     hipGetErrorString(e__));                                       \
      return BSED_ERR_HIP;                                                          \
    }                                                                               \
  } while (0)

#define BSED_HIP(call)                                                              \
  do {                                                                              \
    hipError_t e__ = (call);                                                        \
    if (e__ != hipSuccess) {                                                        \
      bsed_set_error("%s:%d %s: %s", __FILE__, __LINE__, #call,                     \
                     hipGetErrorString(e__));                                       \
      return BSED_ERR_HIP;                                                          \
    }                                                                               \
  } while (0)

// Kernels that need more than 64 KB of dynamic LDS have hipFuncAttributeMaxDynamicSharedMemorySize raised once per
// (kernel, DEVICE): the attribute is per device, so the guard is a per-kernel atomic bitmask indexed by the current
// device, not a process-wide flag (a second GPU in the same process, or two host threads, stay correct; setting
// the attribute twice in a race is harmless).
#include <atomic>
struct BsedLdsOnce { std::atomic<uint64_t> mask{0}; };
static inline hipError_t bsed_max_lds(BsedLdsOnce& once, const void* fn, int bytes = 160 * 1024) {
  int dev = 0;
  hipError_t e = hipGetDevice(&dev);
  if (e != hipSuccess) return e;
  const uint64_t bit = 1ull << (dev & 63);
  if (once.mask.load(std::memory_order_acquire) & bit) return hipSuccess;
  e = hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, bytes);
  if (e == hipSuccess) once.mask.fetch_or(bit, std::memory_order_release);
  return e;
}

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef __bf16 bsed_bf16x2 __attribute__((ext_vector_type(2)));

// Split-fp32 operands: x = hi + lo with hi = bf16(x), lo = bf16(x - hi), both round-to-nearest-even.  gfx950 converts
// two floats to a packed bf16 pair in ONE instruction (v_cvt_pk_bf16_f32); the integer-arithmetic rounding used before
// cost ~11 VALU instructions per element and made the staging loops of the split-fp32 kernels issue-bound.
//   hi / lo : packed pairs, element a in the low half
__device__ __forceinline__ void bsed_split2(float a, float b, uint32_t& hi, uint32_t& lo) {
  const f32x2 v = {a, b};
  hi = __builtin_bit_cast(uint32_t, __builtin_convertvector(v, bsed_bf16x2));
  const f32x2 r = {a - __uint_as_float(hi << 16), b - __uint_as_float(hi & 0xFFFF0000u)};
  lo = __builtin_bit_cast(uint32_t, __builtin_convertvector(r, bsed_bf16x2));
}
typedef float f32x16 __attribute__((ext_vector_type(16)));

#if defined(__HIPCC__)
// ----------------------------------------------------------------------------------------------
// Activation storage of a kernel instance: ABF = 0 fp32, ABF = 1 bf16 (the "bf16" throughput mode of BASELINE
// configs[1-2]: conv outputs, pooled outputs and their gradients are bf16 in HBM; accumulation, BatchNorm statistics,
// master weights and the optimizer stay fp32).  The C ABI keeps `float*` parameter types; with act_bf16 set they
// point to bf16 tensors.
// ----------------------------------------------------------------------------------------------
typedef uint32_t bsed_u32x4 __attribute__((ext_vector_type(4)));
__device__ __forceinline__ float bsed_bf2f(uint32_t h16) { return __uint_as_float(h16 << 16); }
template <int ABF> __device__ __forceinline__ float act_ld(const float* base, size_t i) {
  if (ABF) return bsed_bf2f(reinterpret_cast<const unsigned short*>(base)[i]);
  return base[i];
}
// two-phase form for groups of per-element loads: act_ld_raw issues the load and returns its bits untouched, act_cvt
// turns them into the value LATER.  (With the conversion next to the load hipcc placed every bf16 load's shift -- and
// so an s_waitcnt vmcnt(0) -- directly behind it: one load in flight at a time, the bf16 GLU backward 40 % slower than
// the fp32 one.)  Put __builtin_amdgcn_sched_barrier(0) between the load group and the first act_cvt.
template <int ABF> __device__ __forceinline__ uint32_t act_ld_raw(const float* base, size_t i) {
  if (ABF) return reinterpret_cast<const unsigned short*>(base)[i];
  return __float_as_uint(base[i]);
}
template <int ABF> __device__ __forceinline__ float act_cvt(uint32_t raw) {
  return __uint_as_float(ABF ? raw << 16 : raw);
}
template <int ABF> __device__ __forceinline__ void act_st(float* base, size_t i, float v) {
  if (ABF) reinterpret_cast<__bf16*>(base)[i] = (__bf16)v;   // round to nearest even
  else base[i] = v;
}
// 8 consecutive elements starting at element i (i a multiple of 8): one 16-byte load (bf16) or two (fp32)
template <int ABF> __device__ __forceinline__ void act_ld8(const float* base, size_t i, float* v) {
  if (ABF) {
    const bsed_u32x4 r = *reinterpret_cast<const bsed_u32x4*>(reinterpret_cast<const unsigned short*>(base) + i);
#pragma unroll
    for (int q = 0; q < 4; ++q) { v[2 * q] = __uint_as_float(r[q] << 16); v[2 * q + 1] = __uint_as_float(r[q] & 0xFFFF0000u); }
  } else {
    const f32x4 a = *reinterpret_cast<const f32x4*>(base + i), b = *reinterpret_cast<const f32x4*>(base + i + 4);
    v[0] = a[0]; v[1] = a[1]; v[2] = a[2]; v[3] = a[3]; v[4] = b[0]; v[5] = b[1]; v[6] = b[2]; v[7] = b[3];
  }
}
// 4 consecutive elements starting at element i (i a multiple of 4)
template <int ABF> __device__ __forceinline__ f32x4 act_ld4(const float* base, size_t i) {
  if (ABF) {
    const uint2 r = *reinterpret_cast<const uint2*>(reinterpret_cast<const unsigned short*>(base) + i);
    return f32x4{__uint_as_float(r.x << 16), __uint_as_float(r.x & 0xFFFF0000u), __uint_as_float(r.y << 16),
                 __uint_as_float(r.y & 0xFFFF0000u)};
  }
  return *reinterpret_cast<const f32x4*>(base + i);
}
template <int ABF> __device__ __forceinline__ void act_st4(float* base, size_t i, f32x4 v) {
  if (ABF) {
    const f32x2 a = {v[0], v[1]}, b = {v[2], v[3]};
    uint2 r;
    r.x = __builtin_bit_cast(uint32_t, __builtin_convertvector(a, bsed_bf16x2));
    r.y = __builtin_bit_cast(uint32_t, __builtin_convertvector(b, bsed_bf16x2));
    *reinterpret_cast<uint2*>(reinterpret_cast<unsigned short*>(base) + i) = r;
  } else {
    *reinterpret_cast<f32x4*>(base + i) = v;
  }
}
#endif

static inline int ceil_div(long a, long b) { return (int)((a + b - 1) / b); }

#if defined(__HIPCC__)
// ----------------------------------------------------------------------------------------------
// Philox4x32-10 counter RNG: stateless, keyed on (seed, stream) with a 64-bit element counter, so
// forward and backward regenerate identical dropout masks without storing them.
// ----------------------------------------------------------------------------------------------
__device__ __forceinline__ uint4 philox4x32(uint64_t counter, uint32_t stream, uint64_t seed) {
  uint32_t c0 = (uint32_t)counter, c1 = (uint32_t)(counter >> 32), c2 = stream, c3 = 0x5eed5eedu;
  uint32_t k0 = (uint32_t)seed, k1 = (uint32_t)(seed >> 32);
#pragma unroll
  for (int r = 0; r < 10; ++r) {
    const uint32_t M0 = 0xD2511F53u, M1 = 0xCD9E8D57u;
    uint32_t hi0 = __umulhi(M0, c0), lo0 = M0 * c0;
    uint32_t hi1 = __umulhi(M1, c2), lo1 = M1 * c2;
    uint32_t n0 = hi1 ^ c1 ^ k0, n1 = lo1, n2 = hi0 ^ c3 ^ k1, n3 = lo0;
    c0 = n0; c1 = n1; c2 = n2; c3 = n3;
    k0 += 0x9E3779B9u; k1 += 0xBB67AE85u;
  }
  return make_uint4(c0, c1, c2, c3);
}

// keep-mask for one element: element index e -> 32-bit word (e & 3) of philox(e >> 2)
__device__ __forceinline__ float dropout_scale(uint64_t e, uint32_t stream, uint64_t seed, float p) {
  if (p <= 0.f) return 1.f;
  uint4 r = philox4x32(e >> 2, stream, seed);
  uint32_t w = (e & 3) == 0 ? r.x : (e & 3) == 1 ? r.y : (e & 3) == 2 ? r.z : r.w;
  // uniform in [0,1): keep if u >= p
  float u = (float)(w >> 8) * (1.0f / 16777216.0f);
  return u >= p ? 1.0f / (1.0f - p) : 0.f;
}

__device__ __forceinline__ float sigmoidf_(float x) { return 1.0f / (1.0f + expf(-x)); }
// v_exp_f32 + v_rcp_f32 (1 ulp each): relative error <= ~|x| * 6e-8, used in the streaming epilogues
__device__ __forceinline__ float sigmoid_fast(float x) { return __builtin_amdgcn_rcpf(1.0f + __expf(-x)); }

// ----------------------------------------------------------------------------------------------
// Dropout masks for the fused epilogues: a 32-bit mix hash ("lowbias32") of the element index keyed on
// (seed, layer stream).  ~8 VALU instructions per element instead of ~100 for Philox, stateless, identical
// in forward and backward.  (Philox stays for the Gaussian SNR noise.)
// ----------------------------------------------------------------------------------------------
__device__ __forceinline__ uint32_t mix32(uint32_t x) {
#ifdef BSED_DIAG_NOHASH   // diagnostic builds only (tools/build_variant.sh): what the mask hashes cost; masks are WRONG
  return x;
#endif
  x ^= x >> 16; x *= 0x7feb352dU; x ^= x >> 15; x *= 0x846ca68bU; x ^= x >> 16;
  return x;
}
__device__ __forceinline__ uint32_t drop_key(uint32_t stream, uint64_t seed) {
  return mix32((uint32_t)seed ^ (stream * 0x9E3779B9u)) + mix32((uint32_t)(seed >> 32) ^ 0x85ebca6bu);
}
__device__ __forceinline__ uint32_t drop_threshold(float p) { return (uint32_t)(p * 16777216.0f); }
__device__ __forceinline__ float drop_mul(uint64_t e, uint32_t key, uint32_t thr, float scale) {
  const uint32_t h = mix32((uint32_t)e * 0x9E3779B1u + key + (uint32_t)(e >> 32) * 0x85ebca6bu);
  return (h >> 8) >= thr ? scale : 0.f;
}

// Two keep-multipliers from ONE hash (the first CNN block, whose per-element work is what bounds its kernels): the
// elements e (even) and e + 1 take the low / high 16 bits of the hash of e >> 1 against a 16-bit threshold.  The keep
// probability is 1 - floor(65536 p) / 65536: within 1.6e-5 of 1 - p for any p, exact for p = 0.5.
__device__ __forceinline__ uint32_t drop_threshold16(float p) { return (uint32_t)(p * 65536.0f); }
__device__ __forceinline__ void drop_mul2(uint64_t e_even, uint32_t key, uint32_t thr16, float scale, float& m0,
                                          float& m1) {
  const uint64_t e = e_even >> 1;
  const uint32_t h = mix32((uint32_t)e * 0x9E3779B1u + key + (uint32_t)(e >> 32) * 0x85ebca6bu);
  m0 = (h & 0xFFFFu) >= thr16 ? scale : 0.f;
  m1 = (h >> 16) >= thr16 ? scale : 0.f;
}

// the same two multipliers for element indices below 2^32 (the (e >> 32) term of drop_mul2 is zero there): 32-bit
// index arithmetic only
__device__ __forceinline__ void drop_mul2_32(uint32_t e_even, uint32_t key, uint32_t thr16, float scale, float& m0,
                                             float& m1) {
  const uint32_t h = mix32((e_even >> 1) * 0x9E3779B1u + key);
  m0 = (h & 0xFFFFu) >= thr16 ? scale : 0.f;
  m1 = (h >> 16) >= thr16 ? scale : 0.f;
}

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  return v;
}
__device__ __forceinline__ float wave_max(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o, 64));
  return v;
}
#endif
