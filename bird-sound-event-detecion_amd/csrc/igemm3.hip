// 3x3 convolution forward / data-gradient as an implicit GEMM on the bf16 matrix cores with SPLIT-fp32 operands
// ("bf16x3"): every fp32 operand x is carried as hi = bf16(x), lo = bf16(x - hi) and a product is accumulated as
//     a*b ~= a_hi*b_hi + a_hi*b_lo + a_lo*b_hi          (fp32 accumulate; dropped term a_lo*b_lo ~ 2^-16 relative)
// gfx950 has no xf32/TF32 MFMA; this is its equivalent: ~2e-5 relative error per product (random sign, so ~1e-5 on a
// K = 1152 dot product) at 3 x v_mfma_f32_32x32x16_bf16 = 96 cycles per 16 k instead of 8 x 64 = 512 cycles on
// v_mfma_f32_32x32x2_f32 -- 5.3x fewer matrix-core cycles with the SAME fp32 tensors in HBM.
//
// Same tiling as igemm.hip (128 output positions x BN channels per workgroup, patch + weight slab in LDS).  LDS rows
// are [32 hi | 32 lo | 8 pad] bf16 = 144 B, which keeps the 16-byte fragment reads (lane r, k = 8*(lane>>5)..+7)
// conflict-free: 36 r mod 64 hits 16 distinct 4-bank groups.  Weights are pre-split by bsed_pack_weight3.
#include "bsed_common.h"
#include "../../include/bsed.h"
#include <algorithm>
#include <stdlib.h>

#define I3_THREADS 256
#define I3_M 128
#define I3_KC 32
#define I3_ROW 72  // ushorts per LDS row: 32 hi + 32 lo + 8 pad
// Ablation builds (diagnostic, tools/build_variant.sh <tag> igemm3.hip "-DI3_ABL=<bits>"; results are WRONG by design):
//   1 = no MFMAs (fragments kept live), 2 = no fragment reads from LDS (constant fragments), 4 = no epilogue stores,
//   8 = no weight-slab staging (one slab, no per-tap barriers), 16 = no activation-patch loads after the first chunk
#ifndef I3_ABL
#define I3_ABL 0
#endif

typedef short bf16x8 __attribute__((ext_vector_type(8)));
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));

struct Igemm3Params {
  BsedIgemmDesc d;
  int PW, PH, PP, lgTW, b_off, pw_magic;
};

__device__ __forceinline__ unsigned short f2bf(float x) {
  const uint32_t u = __float_as_uint(x);
  return (unsigned short)((u + 0x7FFFu + ((u >> 16) & 1u)) >> 16);
}
__device__ __forceinline__ float bf2f(unsigned short h) { return __uint_as_float((uint32_t)h << 16); }
__device__ __forceinline__ int crow3(int r, int lh) { return (r & 3) + 8 * (r >> 2) + 4 * lh; }

// RB = 32-row blocks per wave: 1 -> a workgroup covers 128 positions, 2 -> 256 (TH*TW = 256).  With RB = 2 every B
// fragment read from LDS feeds two MFMA row blocks: 24 ds_read_b128 per 48 MFMAs instead of 20 per 24 -- the LDS pipe
// (one per CU, shared by all resident waves) was the co-limiter of the RB = 1 kernel at ~30 % of the MFMA peak -- and a
// weight slab is staged once per 256 positions instead of once per 128.
template <int BN, int STATS, int RB, int PV>
// (waves_per_eu caps the register budget the allocator aims for: without it the prefetch registers were spilled to
// scratch right after their loads, which serialised the loads again)
// (BN = 128, RB = 1: given "1 to 3" the allocator settled at 228 registers = two workgroups per CU although the
// kernel fits 152 without a spill; with 38 % of a workgroup's life outside its MFMA loop -- in-kernel stamps -- the
// third resident workgroup is worth 7 %, so there the minimum is pinned to 3 as well)
__global__ __launch_bounds__(I3_THREADS)
// (BN <= 64: pinned to FOUR waves per SIMD, i.e. 128 registers and a fourth resident workgroup: 0.446 / 0.335 / 0.312 ms ->
//  0.41 / 0.295 / 0.271 on the 64- and 32-channel layers; A/B: -DI3_WPE_SMALL=3)
#ifndef I3_WPE_SMALL
#define I3_WPE_SMALL 4
#endif
__attribute__((amdgpu_waves_per_eu(RB == 2 ? 1 : ((BN == 128 && PV <= 9) ? 3 : (BN <= 64 && I3_WPE_SMALL > 3 ? I3_WPE_SMALL : 1)),
                                   RB == 2 ? 2 : (BN <= 64 ? I3_WPE_SMALL : 3))))  // (PV = 12 would spill at 3)
void igemm3_kernel(const Igemm3Params P) {
  constexpr int NT = BN / 32;
  const BsedIgemmDesc& p = P.d;
  extern __shared__ __align__(16) unsigned short smem3[];
  unsigned short* As = smem3;             // [PP][72]
  unsigned short* Bs = smem3 + P.b_off;   // [BN][72]
  const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63, li = lane & 31, lh = lane >> 5;
  int tile = blockIdx.x;
  const int tw_i = tile % p.tilesW; tile /= p.tilesW;
  const int th_i = tile % p.tilesH;
  const int nb = tile / p.tilesH;
  const int th0 = th_i * p.TH, tw0 = tw_i * p.TW;
  const int n0 = blockIdx.y * BN;
  const int PW = P.PW;
  int abase[RB];
#pragma unroll
  for (int rb = 0; rb < RB; ++rb) {
    const int m = (wave * RB + rb) * 32 + li;
    abase[rb] = (((m >> P.lgTW) + p.hh) * PW + (m & (p.TW - 1)) + p.hw) * I3_ROW + 8 * lh;
  }

  f32x16 acc[RB][NT];
#pragma unroll
  for (int rb = 0; rb < RB; ++rb)
#pragma unroll
    for (int j = 0; j < NT; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[rb][j][r] = 0.f;

  const float* inb = p.in + (size_t)nb * p.H * p.W * p.in_pitch;
  const int nchunks = p.CIN / I3_KC;
  // Activation patch of one 32-channel chunk: PV float4 per thread.  The patch of chunk c+1 is fetched into REGISTERS
  // during the taps of chunk c (its HBM latency was ~1/3 of the kernel time when exposed at every chunk boundary) and
  // split into bf16 hi / lo when it is written to LDS.  Element e = tid + u*256 -> (position e/8, channels 4*(e%8)..).
  const int a_total = P.PP * (I3_KC / 4);
  f32x4 pv[PV];
  int poff[PV];  // element offset of this thread's u-th float4 inside the image (channel chunk 0), -1 = zero padding
#pragma unroll
  for (int u = 0; u < PV; ++u) {
    const int e = tid + u * I3_THREADS;
    const int c4 = e & 7, pos = e >> 3;
    const int pr = (pos * P.pw_magic) >> 20, pc = pos - pr * PW;
    const int gh = th0 - p.hh + pr, gw = tw0 - p.hw + pc;
    poff[u] = (e < a_total && gh >= 0 && gh < p.H && gw >= 0 && gw < p.W) ? (gh * p.W + gw) * p.in_pitch + 4 * c4 : -1;
  }
#pragma unroll
  for (int u = 0; u < PV; ++u)
    pv[u] = poff[u] >= 0 ? *reinterpret_cast<const f32x4*>(inb + poff[u]) : f32x4{0.f, 0.f, 0.f, 0.f};
  for (int ch = 0; ch < nchunks; ++ch) {
    __syncthreads();  // every wave is done with the previous chunk's patch and slab
#pragma unroll
    for (int u = 0; u < PV; ++u) {
      const int e = tid + u * I3_THREADS;
      if (e < a_total) {
        uint32_t h01, l01, h23, l23;
        bsed_split2(pv[u][0], pv[u][1], h01, l01);
        bsed_split2(pv[u][2], pv[u][3], h23, l23);
        unsigned short* dst = As + (e >> 3) * I3_ROW + 4 * (e & 7);
        *reinterpret_cast<uint2*>(dst) = make_uint2(h01, h23);
        *reinterpret_cast<uint2*>(dst + 32) = make_uint2(l01, l23);
      }
    }
    // weight slab [BN][64] bf16 (hi | lo), pre-split: 128 B per output channel, NT uint4 per thread.  The slab of tap
    // t+1 is fetched into REGISTERS before the MFMAs of tap t and written to the (single) LDS buffer after the next
    // barrier, so its L2 latency hides behind 24 MFMAs per wave without costing LDS (three workgroups stay resident).
    u32x4 pre[NT];  // (a native vector type: HIP's uint4 struct array was left in scratch memory)
    // this thread's NT uint4 of a slab sit at a fixed offset from the slab base (8 uint4 = 128 B per output channel)
    const u32x4* wthr = reinterpret_cast<const u32x4*>(p.w) + ((size_t)ch * p.NP + n0) * 8 + tid;
    const size_t tap_stride = (size_t)nchunks * p.NP * 8;  // uint4 per tap
#pragma unroll
    for (int u = 0; u < NT; ++u) pre[u] = wthr[u * I3_THREADS];
    for (int tap = 0; tap < p.ntaps; ++tap) {
      if (!(I3_ABL & 8) || tap == 0) {
      if (tap > 0) __syncthreads();  // every wave is done reading the previous slab
#pragma unroll
      for (int u = 0; u < NT; ++u) {
        const int e = tid + u * I3_THREADS;
        *reinterpret_cast<u32x4*>(Bs + (e >> 3) * I3_ROW + (e & 7) * 8) = pre[u];
      }
      __syncthreads();
      }
      if (tap == 0 && ch + 1 < nchunks && !(I3_ABL & 16)) {  // next chunk's patch: in flight during this chunk's taps
#pragma unroll
        for (int u = 0; u < PV; ++u)
          pv[u] = poff[u] >= 0 ? *reinterpret_cast<const f32x4*>(inb + poff[u] + (ch + 1) * I3_KC) : f32x4{0.f, 0.f, 0.f, 0.f};
      }
      if (!(I3_ABL & 8)) {  // next tap's slab (the last tap re-reads its own: no branch around the loads)
        const u32x4* wn = wthr + (size_t)min(tap + 1, p.ntaps - 1) * tap_stride;
#pragma unroll
        for (int u = 0; u < NT; ++u) pre[u] = wn[u * I3_THREADS];
      }
      const int toff = (p.dh[tap] * PW + p.dw[tap]) * I3_ROW;
      const unsigned short* brow = Bs + li * I3_ROW + 8 * lh;
      // every operand fragment of the tap is fetched from LDS first (20 ds_read_b128 at RB = 1), then the 24 * RB
      // MFMAs run back to back: one LDS wait per tap instead of one per accumulator tile (PMC: the waves of the
      // read-then-multiply loop were parked 50 % of their cycles with the MFMA pipe 31 % busy)
      bf16x8 a_hi[2][RB], a_lo[2][RB], b_hi[2][NT], b_lo[2][NT];
#pragma unroll
      for (int kk = 0; kk < 2; ++kk) {
#pragma unroll
        for (int rb = 0; rb < RB; ++rb) {
          if (I3_ABL & 2) {
            a_hi[kk][rb] = bf16x8{(short)toff, 1, 2, 3, 4, 5, 6, (short)tid}; a_lo[kk][rb] = a_hi[kk][rb];
          } else {
          a_hi[kk][rb] = *reinterpret_cast<const bf16x8*>(As + abase[rb] + toff + 16 * kk);
          a_lo[kk][rb] = *reinterpret_cast<const bf16x8*>(As + abase[rb] + toff + 32 + 16 * kk);
          }
        }
#pragma unroll
        for (int j = 0; j < NT; ++j) {
          if (I3_ABL & 2) {
            b_hi[kk][j] = bf16x8{(short)tap, 1, 2, 3, 4, 5, 6, (short)(tid + j)}; b_lo[kk][j] = b_hi[kk][j];
          } else {
          b_hi[kk][j] = *reinterpret_cast<const bf16x8*>(brow + 32 * j * I3_ROW + 16 * kk);
          b_lo[kk][j] = *reinterpret_cast<const bf16x8*>(brow + 32 * j * I3_ROW + 32 + 16 * kk);
          }
        }
      }
      // independent accumulators are interleaved so that consecutive MFMAs never depend on each other
      if (I3_ABL & 1) {   // keep the fragments live without multiplying
#pragma unroll
        for (int kk = 0; kk < 2; ++kk) {
#pragma unroll
          for (int rb = 0; rb < RB; ++rb) asm volatile("" :: "v"(a_hi[kk][rb]), "v"(a_lo[kk][rb]));
#pragma unroll
          for (int j = 0; j < NT; ++j) asm volatile("" :: "v"(b_hi[kk][j]), "v"(b_lo[kk][j]));
        }
      } else
#pragma unroll
      for (int kk = 0; kk < 2; ++kk) {
#pragma unroll
        for (int rb = 0; rb < RB; ++rb)
#pragma unroll
          for (int j = 0; j < NT; ++j) acc[rb][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a_lo[kk][rb], b_hi[kk][j], acc[rb][j], 0, 0, 0);
#pragma unroll
        for (int rb = 0; rb < RB; ++rb)
#pragma unroll
          for (int j = 0; j < NT; ++j) acc[rb][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a_hi[kk][rb], b_lo[kk][j], acc[rb][j], 0, 0, 0);
#pragma unroll
        for (int rb = 0; rb < RB; ++rb)
#pragma unroll
          for (int j = 0; j < NT; ++j) acc[rb][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a_hi[kk][rb], b_hi[kk][j], acc[rb][j], 0, 0, 0);
      }
    }
  }

  // ---- epilogue: + bias, store, optional BatchNorm partial sums (same as igemm.hip PLAIN / STATS)
  float s0[NT], s1[NT], bias[NT];
  bool nok[NT];
#pragma unroll
  for (int j = 0; j < NT; ++j) {
    s0[j] = 0.f; s1[j] = 0.f;
    const int n = n0 + 32 * j + li;
    nok[j] = n < p.N;
    bias[j] = (p.bias && nok[j]) ? p.bias[n] : 0.f;
  }
  const int nbase = n0 + li;
#pragma unroll
  for (int rb = 0; rb < RB; ++rb)
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int mm = (wave * RB + rb) * 32 + crow3(r, lh);
      const int gh = th0 + (mm >> P.lgTW), gw = tw0 + (mm & (p.TW - 1));
      const bool pok = gh < (p.valid_h > 0 ? p.valid_h : p.H) && gw < (p.valid_w > 0 ? p.valid_w : p.W);
      float* orow = p.out + (((size_t)nb * p.H + gh) * p.W + gw) * p.out_pitch + nbase;
#pragma unroll
      for (int j = 0; j < NT; ++j) {
        if (pok && nok[j]) {
          const float v = acc[rb][j][r] + bias[j];
          if (!(I3_ABL & 4) || v == 123.456f) orow[32 * j] = v;
          if (STATS) { s0[j] += v; s1[j] = fmaf(v, v, s1[j]); }
        }
      }
    }
  if (STATS) {
    __syncthreads();
    float* red = reinterpret_cast<float*>(smem3);  // [4][2][BN]
#pragma unroll
    for (int j = 0; j < NT; ++j) {
      const float a = s0[j] + __shfl_xor(s0[j], 32, 64);
      const float b = s1[j] + __shfl_xor(s1[j], 32, 64);
      if (lh == 0) {
        red[(wave * 2 + 0) * BN + 32 * j + li] = a;
        red[(wave * 2 + 1) * BN + 32 * j + li] = b;
      }
    }
    __syncthreads();
    if (tid < 2 * BN) {
      const int which = tid / BN, n = tid % BN;
      if (n0 + n < p.N) {
        const float s = red[(0 * 2 + which) * BN + n] + red[(1 * 2 + which) * BN + n] + red[(2 * 2 + which) * BN + n] +
                        red[(3 * 2 + which) * BN + n];
        p.stats[((size_t)blockIdx.x * 2 + which) * p.N + n0 + n] = s;
      }
    }
  }
}

// w3[tap][chunk][n][0..31] = bf16 hi, [32..63] = bf16 lo of src[tap*s_tap + (chunk*32+k)*s_k + n*s_n]; zero for n >= N
__device__ __forceinline__ void pack3_elem(const float* __restrict__ src, unsigned short* __restrict__ dst, long e,
                                           int nchunks, int N, int NP, long s_tap, long s_k, long s_n) {
  const int k = (int)(e & 31);
  long r = e >> 5;
  const int n = (int)(r % NP); r /= NP;
  const int ch = (int)(r % nchunks);
  const int tap = (int)(r / nchunks);
  const float v = n < N ? src[tap * s_tap + (long)(ch * 32 + k) * s_k + n * s_n] : 0.f;
  const unsigned short hi = f2bf(v);
  const unsigned short lo = f2bf(v - bf2f(hi));
  unsigned short* d = dst + (((long)tap * nchunks + ch) * NP + n) * 64;
  d[k] = hi;
  d[32 + k] = lo;
}

__global__ void pack_weight3_kernel(const float* __restrict__ src, unsigned short* __restrict__ dst, int ntaps, int K,
                                    int N, int NP, long s_tap, long s_k, long s_n) {
  const int nchunks = K / 32;
  const long total = (long)ntaps * nchunks * NP * 32;
  for (long e = (long)blockIdx.x * blockDim.x + threadIdx.x; e < total; e += (long)gridDim.x * blockDim.x)
    pack3_elem(src, dst, e, nchunks, N, NP, s_tap, s_k, s_n);
}

// fragment-layout self test of v_mfma_f32_32x32x16_bf16: C(32,32) = A(32,K) B(K,32) with bf16x3 split operands
__global__ void mfma_bf16x3_selftest_kernel(const float* A, const float* B, float* C, int K) {
  const int lane = threadIdx.x & 63, li = lane & 31, lh = lane >> 5;
  f32x16 acc = {0};
  for (int k0 = 0; k0 < K; k0 += 16) {
    bf16x8 ah, al, bh, bl;
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      const float a = A[li * K + k0 + 8 * lh + j];         // A[row = lane&31][k = 8*(lane>>5) + j]
      const float b = B[(k0 + 8 * lh + j) * 32 + li];      // B[k = 8*(lane>>5) + j][col = lane&31]
      const unsigned short ahi = f2bf(a), bhi = f2bf(b);
      ah[j] = (short)ahi; al[j] = (short)f2bf(a - bf2f(ahi));
      bh[j] = (short)bhi; bl[j] = (short)f2bf(b - bf2f(bhi));
    }
    acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(al, bh, acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah, bl, acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah, bh, acc, 0, 0, 0);
  }
#pragma unroll
  for (int r = 0; r < 16; ++r) C[crow3(r, lh) * 32 + li] = acc[r];
}

extern "C" int bsed_selftest_mfma_bf16x3(const float* A, const float* B, float* C, int K, void* stream) {
  BSED_CHECK_ARG(A && B && C && K > 0 && K % 16 == 0, "bsed_selftest_mfma_bf16x3: bad argument");
  hipLaunchKernelGGL(mfma_bf16x3_selftest_kernel, dim3(1), dim3(64), 0, (hipStream_t)stream, A, B, C, K);
  BSED_LAUNCH_CHECK();
  return BSED_OK;
}

extern "C" int bsed_pack_weight3(const float* src, void* dst, int ntaps, int K, int N, int NP, long s_tap, long s_k,
                                 long s_n, void* stream) {
  BSED_CHECK_ARG(src && dst && ntaps > 0 && K > 0 && K % 32 == 0 && N > 0 && NP >= N && NP % 32 == 0,
                 "bsed_pack_weight3: K must be a multiple of 32, NP a multiple of 32 >= N");
  const long total = (long)ntaps * K * NP;
  hipLaunchKernelGGL(pack_weight3_kernel, dim3((unsigned)std::min<long>(ceil_div(total, 256), 4096)), dim3(256), 0,
                     (hipStream_t)stream, src, (unsigned short*)dst, ntaps, K, N, NP, s_tap, s_k, s_n);
  BSED_LAUNCH_CHECK();
  return BSED_OK;
}

// ---------------------------------------------------------------------------------------------------------------
// CIN = 16 (the second convolution of the network, 16 -> 32 channels on the 432 x 64 map): one MFMA K step per tap.
// With 3 MFMAs per tap the two-barriers-per-tap pipeline above is all overhead, so this variant keeps the pre-split
// weights of ALL taps in LDS (fragment order, 18 KB per 32 output channels, loaded once per persistent workgroup) and
// only the activation patch moves per tile: [prefetched registers -> LDS] -> barrier -> 27 MFMAs per wave -> epilogue.
// BatchNorm partial sums accumulate in registers over the tiles of a workgroup (one partial row per workgroup).
// (KS = CIN / 16 K steps per tap: 1 for the 16 -> 32 convolution, 2 for the data gradients of 32-channel layers)

// table[jn][tap][kk][hi|lo][lane][8]: lane (li, lh) holds k = 16*kk + 8*lh + q of output channel n = 32*jn + li
__device__ __forceinline__ void pack3s_elem(const float* __restrict__ src, unsigned short* __restrict__ dst, long e,
                                            int ntaps, int KS, int N, long s_tap, long s_k, long s_n) {
  const int lane = (int)(e & 63), li = lane & 31, lh = lane >> 5;
  long r = e >> 6;
  const int kk = (int)(r % KS); r /= KS;
  const int tap = (int)(r % ntaps), jn = (int)(r / ntaps);
  const int n = 32 * jn + li;
  unsigned short* d = dst + (((((long)jn * ntaps + tap) * KS + kk) * 2) * 64 + lane) * 8;
#pragma unroll
  for (int q = 0; q < 8; ++q) {
    const float v = n < N ? src[tap * s_tap + (long)(16 * kk + 8 * lh + q) * s_k + n * s_n] : 0.f;
    const unsigned short hi = f2bf(v);
    d[q] = hi;
    d[64 * 8 + q] = f2bf(v - bf2f(hi));
  }
}

__global__ void pack_weight3s_kernel(const float* __restrict__ src, unsigned short* __restrict__ dst, int ntaps, int KS,
                                     int N, int NP, long s_tap, long s_k, long s_n) {
  const long total = (long)(NP / 32) * ntaps * KS * 64;
  for (long e = (long)blockIdx.x * blockDim.x + threadIdx.x; e < total; e += (long)gridDim.x * blockDim.x)
    pack3s_elem(src, dst, e, ntaps, KS, N, s_tap, s_k, s_n);
}

// Every weight re-layout of a train step in ONE launch (the packs are 5-7 us kernels on the forward's critical path:
// fourteen launches cost ~0.19 ms per step in event time).  Same per-element code as the two kernels above: same bits.
struct PkJob {
  const float* src; unsigned short* dst;
  int kind, ntaps, KS, N, NP, wg0;   // kind 0: pack_weight3 layout (KS = K / 32 chunks), 1: pack_weight3s (KS = K / 16)
  long total, s_tap, s_k, s_n;
};
struct PkBatch { PkJob j[BSED_PACK_MAX_JOBS]; int njobs; };

__global__ __launch_bounds__(256) void pack_weights_batch_kernel(const PkBatch Bt) {
  int ji = 0;
  for (int k = 1; k < Bt.njobs; ++k)
    if ((int)blockIdx.x >= Bt.j[k].wg0) ji = k;
  const PkJob& J = Bt.j[ji];
  const long e = (long)((int)blockIdx.x - J.wg0) * 256 + threadIdx.x;
  if (e >= J.total) return;
  if (J.kind == 0) pack3_elem(J.src, J.dst, e, J.KS, J.N, J.NP, J.s_tap, J.s_k, J.s_n);
  else pack3s_elem(J.src, J.dst, e, J.ntaps, J.KS, J.N, J.s_tap, J.s_k, J.s_n);
}

// NV = output channels a workgroup really has (32, or 16 for N <= 16): with 16 the plain epilogue would store 64-byte
// pieces from half the lanes, 16 instructions per tile; instead the tile goes through a wave-private LDS image and
// leaves as float4 rows, 1 KB of consecutive addresses per instruction (0.59 -> 0.45 ms on the 32 -> 16 channel dgrad).
#define I3S_EROW 16  // floats per position row of that image (no pad: the two rows a store instruction touches are 4 rows
                     // apart = the same banks, a 2-way conflict that a 4-byte store absorbs; 8 KB instead of 10 per workgroup,
                     // which -- with the half-width weight table below -- lets a THIRD workgroup share the CU's LDS)

// ABF = 1 ("bf16" mode): `in` and `out` are bf16 tensors, ONE MFMA per product (the hi halves of the same weight
// table), patch rows CIN bf16 + 8 pad (48 B / 80 B: odd x 16 B), 16-byte patch pieces of 8 channels
template <int STATS, int NTAPS, int KS, int NV, int ABF>
__global__ __launch_bounds__(I3_THREADS) void igemm3s_kernel(const Igemm3Params P) {
  constexpr int CINK = 16 * KS;
  constexpr int I3S_ROW = (ABF ? CINK : 2 * CINK) + 8;  // ushorts per patch row: CIN hi | CIN lo | 8 pad (80 B / 144 B: odd x 16 B)
  constexpr int C4 = ABF ? CINK / 8 : CINK / 4;   // 16-byte pieces per patch position (4 fp32 / 8 bf16 channels)
  constexpr int PV = ABF ? (KS == 1 ? 2 : 3) : (KS == 1 ? 4 : 6);    // patch pieces per thread: 256 x 4 / 192 x 8 patch elements
  const BsedIgemmDesc& p = P.d;
  extern __shared__ __align__(16) unsigned short smem3[];
  // NV == 16: output channels 16..31 of the 32-wide MFMA tile do not exist, their weight fragments are never stored:
  // the table keeps the 32 lanes (li < 16, lh) of every fragment and lanes li >= 16 read the fragment of li - 16
  constexpr int WL = NV == 16 ? 32 : 64;
  u32x4* Wf = reinterpret_cast<u32x4*>(smem3);               // [NTAPS][KS][2][WL]
  unsigned short* As = smem3 + NTAPS * KS * 2 * WL * 8;      // [PP][I3S_ROW]
  const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63, li = lane & 31, lh = lane >> 5;
  float* Es = reinterpret_cast<float*>(As + P.PP * I3S_ROW) + wave * 32 * I3S_EROW;  // NV == 16 only
  const int jn = blockIdx.y, n0 = 32 * jn;
  const int PW = P.PW;
  for (int i = tid; i < NTAPS * KS * 2 * WL; i += I3_THREADS) {
    const int fr = i / WL, l = i % WL;
    const int src = NV == 16 ? (l & 15) + 32 * (l >> 4) : l;
    Wf[i] = reinterpret_cast<const u32x4*>(p.w)[(size_t)jn * NTAPS * KS * 2 * 64 + fr * 64 + src];
  }
  const int wlane = NV == 16 ? (li & 15) + 16 * lh : lane;
  const int m = wave * 32 + li;
  const int abase = (((m >> P.lgTW) + p.hh) * PW + (m & (p.TW - 1)) + p.hw) * I3S_ROW + 8 * lh;
  int toff[NTAPS];
#pragma unroll
  for (int t = 0; t < NTAPS; ++t) toff[t] = (p.dh[t] * PW + p.dw[t]) * I3S_ROW;
  const int n = n0 + li;
  const bool nok = n < p.N;
  float bias = (p.bias && nok) ? p.bias[n] : 0.f;
  // consumed HERE: otherwise the compiler's wait for this load sits in front of its first use inside the tile loop and
  // runs every iteration -- as vmcnt(0), i.e. as a wait for the patch prefetch as well
  asm volatile("" : "+v"(bias));
  float s0 = 0.f, s1 = 0.f;
  const int a_total = P.PP * C4;
  const int ntiles = p.NB * p.tilesH * p.tilesW;

  // patch element e = tid + u*256 -> (position e/C4, channels 4*(e%C4)..): tile-independent part of the address
  int ppr[PV], ppc[PV];
#pragma unroll
  for (int u = 0; u < PV; ++u) {
    const int e = tid + u * I3_THREADS, pos = e / C4;
    ppr[u] = (pos * P.pw_magic) >> 20;
    ppc[u] = pos - ppr[u] * PW;
  }
  // The patch prefetch is issued by inline assembly and waited for by hand.  Its loads (tile t + 1) are older than the
  // epilogue stores of tile t, so the wait in front of the next LDS write may leave those stores in flight -- s_waitcnt
  // vmcnt(#stores) -- but hipcc's own wait insertion gives vmcnt(0) at a loop head whatever the body looks like, and then
  // every tile starts by draining its predecessor's stores (1-2 k cycles with three workgroups per CU to cover for it).
  // The compiler does not know these are memory instructions: nothing may touch pv between issue() and patch_wait() --
  // tools/asm_load_check.py verifies that on the listing.  Loads are branch-free: positions outside the map read a
  // clamped address and become zeros at the LDS write.
  // HANDWAIT only in the 32-channel-output form: in the 16-channel form (its epilogue goes through an LDS image and
  // needs more registers) the allocator moved in-flight registers around -- caught by the listing check, not by a test --
  // and the instance did not gain anyway (its busiest unit is the LDS array); it keeps compiler-tracked loads.
  constexpr bool HANDWAIT = NV == 32;
  u32x4 pv[PV];
  auto issue = [&](int tile) {
    const int tw_i = tile % p.tilesW; const int r1 = tile / p.tilesW;
    const int th_i = r1 % p.tilesH, nb = r1 / p.tilesH;
    const char* inb = reinterpret_cast<const char*>(p.in) + (size_t)nb * p.H * p.W * p.in_pitch * (ABF ? 2 : 4);
#pragma unroll
    for (int u = 0; u < PV; ++u) {
      const int e = tid + u * I3_THREADS;
      if (HANDWAIT) {
        const int gh = min(max(th_i * p.TH - p.hh + ppr[u], 0), p.H - 1), gw = min(max(tw_i * p.TW - p.hw + ppc[u], 0), p.W - 1);
        const char* a = inb + (((size_t)gh * p.W + gw) * p.in_pitch * (ABF ? 2 : 4)) + 16 * (e % C4);
        asm volatile("global_load_dwordx4 %0, %1, off" : "=v"(pv[u]) : "v"(a) : "memory");
      } else {   // tracked by the compiler: its own waits apply
        const int gh = th_i * p.TH - p.hh + ppr[u], gw = tw_i * p.TW - p.hw + ppc[u];
        const bool ok = e < a_total && gh >= 0 && gh < p.H && gw >= 0 && gw < p.W;
        pv[u] = ok ? *reinterpret_cast<const u32x4*>(inb + (((size_t)gh * p.W + gw) * p.in_pitch * (ABF ? 2 : 4)) + 16 * (e % C4))
                   : u32x4{0u, 0u, 0u, 0u};
      }
    }
  };
  // all but the `keep` youngest vector-memory operations of the wave are done
  auto patch_wait = [&](bool counted) {
    if (!HANDWAIT) return;
    constexpr int NST = 16;   // epilogue stores per lane and tile when nothing is masked
    if (!counted) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    else asm volatile("s_waitcnt vmcnt(16)" ::: "memory");
#pragma unroll
    for (int u = 0; u < PV; ++u) asm volatile("" : "+v"(pv[u]));
  };
  // every store of every tile of this workgroup is unmasked: whole tiles and whole channel groups
  const bool allfull = p.H % p.TH == 0 && n0 + NV <= p.N;
  // The wait sits at the END of a tile's iteration (behind its stores, in the same branch arm), so the registers are valid
  // data again before control reaches the loop's back edge: register copies the allocator places there are harmless.
  if ((int)blockIdx.x < ntiles) { issue(blockIdx.x); patch_wait(false); }

  // WREG (the 32 -> 16 channel data gradient, 18 fragment pairs): the hi weight fragments live in registers for the whole
  // kernel.  From LDS, every MFMA triple costs four 16-byte wave reads (two activation, two weight fragments): with
  // three workgroups per CU the LDS array was the busiest unit of this instance (0.65, matrix pipe 0.50).
  constexpr bool WREG = NV == 16 && KS == 2 && NTAPS == 9 && !STATS;
  // ... and that instance contracts on v_mfma_f32_16x16x32_bf16: a 16-column tile is exactly its output (the 32 x 32 form
  // computes 16 columns that do not exist), and K = 32 is exactly its input channels: per tap 2 row tiles x 3 MFMAs of
  // ~16 cycles instead of 2 k steps x 3 MFMAs of 32 cycles, one weight fragment per tap instead of two (9 registers x 4
  // for the hi parts).  Lane (i = lane & 15, kg = lane >> 4): row i of the tile, channels 8 kg .. 8 kg + 7; the weight
  // table's 32 x 32 x 16 fragments (k step kg >> 1, half kg & 1) hold just those values.
  constexpr bool M16 = WREG;
  const int li16 = lane & 15, kg = lane >> 4;
  int abase16[2];
#pragma unroll
  for (int rt = 0; rt < 2; ++rt) {
    const int m16 = wave * 32 + 16 * rt + li16;
    abase16[rt] = (((m16 >> P.lgTW) + p.hh) * PW + (m16 & (p.TW - 1)) + p.hw) * I3S_ROW + 8 * kg;
  }
  const int wfrag16 = ((kg >> 1) * 2) * WL + li16 + 16 * (kg & 1);   // + (t * KS * 2 + hl) * WL
  // (fp32 activations: only the hi fragments in registers)
  u32x4 wreg[WREG ? NTAPS : 1];
  if (WREG) {
    __syncthreads();   // Wf is complete
#pragma unroll
    for (int t = 0; t < NTAPS; ++t) wreg[t] = Wf[(t * KS * 2 + 0) * WL + wfrag16];
  }
  const float bias16 = (M16 && p.bias && n0 + li16 < p.N) ? p.bias[n0 + li16] : 0.f;
  for (int tile = blockIdx.x; tile < ntiles; tile += gridDim.x) {
    const int tw_i = tile % p.tilesW; const int r1 = tile / p.tilesW;
    const int th_i = r1 % p.tilesH, nb = r1 / p.tilesH;
    const int th0 = th_i * p.TH, tw0 = tw_i * p.TW;
    __syncthreads();  // every wave is done reading the previous patch (and, first time, Wf is complete)
#pragma unroll
    for (int u = 0; u < PV; ++u) {
      const int e = tid + u * I3_THREADS;
      if (e < a_total) {
        const int gh = th0 - p.hh + ppr[u], gw = tw0 - p.hw + ppc[u];
        const bool inside = !HANDWAIT || (gh >= 0 && gh < p.H && gw >= 0 && gw < p.W);   // (tracked loads come zeroed)
        const u32x4 pvu = inside ? pv[u] : u32x4{0u, 0u, 0u, 0u};
        if (ABF) {
          *reinterpret_cast<u32x4*>(As + (e / C4) * I3S_ROW + 8 * (e % C4)) = pvu;
        } else {
          uint32_t h01, l01, h23, l23;
          const f32x4 v = __builtin_bit_cast(f32x4, pvu);
          bsed_split2(v[0], v[1], h01, l01);
          bsed_split2(v[2], v[3], h23, l23);
          unsigned short* dst = As + (e / C4) * I3S_ROW + 4 * (e % C4);
          *reinterpret_cast<uint2*>(dst) = make_uint2(h01, h23);
          *reinterpret_cast<uint2*>(dst + CINK) = make_uint2(l01, l23);
        }
      }
    }
    __syncthreads();
    // next tile's patch: in flight during the MFMAs (hand-waited form: the last tile requests a dummy, no branch)
    if (HANDWAIT) issue(min(tile + (int)gridDim.x, ntiles - 1));
    else if (tile + (int)gridDim.x < ntiles) issue(tile + gridDim.x);

    f32x16 acc;
    f32x4 acc16[2] = {{0.f, 0.f, 0.f, 0.f}, {0.f, 0.f, 0.f, 0.f}};
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[r] = 0.f;
    if constexpr (M16) {
#pragma unroll
      for (int t = 0; t < NTAPS; ++t) {
        const bf16x8 b_hi = __builtin_bit_cast(bf16x8, wreg[t]);
        bf16x8 a_hi[2], a_lo[2];
#pragma unroll
        for (int rt = 0; rt < 2; ++rt) {
          a_hi[rt] = *reinterpret_cast<const bf16x8*>(As + abase16[rt] + toff[t]);
          if (!ABF) a_lo[rt] = *reinterpret_cast<const bf16x8*>(As + abase16[rt] + toff[t] + CINK);
        }
        if (!ABF) {
          const bf16x8 b_lo = __builtin_bit_cast(bf16x8, Wf[(t * KS * 2 + 1) * WL + wfrag16]);
#pragma unroll
          for (int rt = 0; rt < 2; ++rt) acc16[rt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a_lo[rt], b_hi, acc16[rt], 0, 0, 0);
#pragma unroll
          for (int rt = 0; rt < 2; ++rt) acc16[rt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a_hi[rt], b_lo, acc16[rt], 0, 0, 0);
        }
#pragma unroll
        for (int rt = 0; rt < 2; ++rt) acc16[rt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a_hi[rt], b_hi, acc16[rt], 0, 0, 0);
      }
    } else
#pragma unroll
    for (int kk = 0; kk < KS; ++kk) {
      bf16x8 a_hi[NTAPS], a_lo[NTAPS];
#pragma unroll
      for (int t = 0; t < NTAPS; ++t) {
        a_hi[t] = *reinterpret_cast<const bf16x8*>(As + abase + toff[t] + 16 * kk);
        if (!ABF) a_lo[t] = *reinterpret_cast<const bf16x8*>(As + abase + toff[t] + CINK + 16 * kk);
      }
#pragma unroll
      for (int t = 0; t < NTAPS; ++t) {
        const bf16x8 b_hi = __builtin_bit_cast(bf16x8, Wf[((t * KS + kk) * 2 + 0) * WL + wlane]);
        if (!ABF) {
          const bf16x8 b_lo = __builtin_bit_cast(bf16x8, Wf[((t * KS + kk) * 2 + 1) * WL + wlane]);
          acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a_lo[t], b_hi, acc, 0, 0, 0);
          acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a_hi[t], b_lo, acc, 0, 0, 0);
        }
        acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a_hi[t], b_hi, acc, 0, 0, 0);
      }
    }
    if (NV == 16) {
      if constexpr (M16) {
        // result fragment of the 16 x 16 tile: column li16, rows 4 kg + r
#pragma unroll
        for (int rt = 0; rt < 2; ++rt)
#pragma unroll
          for (int r = 0; r < 4; ++r) Es[(16 * rt + 4 * kg + r) * I3S_EROW + li16] = acc16[rt][r] + bias16;
      } else
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int row = crow3(r, lh);
        const float v = acc[r] + bias;
        if (li < 16) Es[row * I3S_EROW + li] = v;
        if (STATS) {
          const int mm = wave * 32 + row;
          if (th0 + (mm >> P.lgTW) < p.H && nok) { s0 += v; s1 = fmaf(v, v, s1); }
        }
      }
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");  // wave-private image: stores before the wide reads below
#pragma unroll
      for (int k = 0; k < 2; ++k) {
        const int e = lane + 64 * k, pos = e >> 2, q = e & 3;
        const int mm = wave * 32 + pos;
        const int gh = th0 + (mm >> P.lgTW), gw = tw0 + (mm & (p.TW - 1));
        const f32x4 v = *reinterpret_cast<const f32x4*>(Es + pos * I3S_EROW + 4 * q);
        if (gh < p.H && n0 + 4 * q < p.N)
          act_st4<ABF>(p.out, (((size_t)nb * p.H + gh) * p.W + gw) * p.out_pitch + n0 + 4 * q, v);
      }
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");  // reads done before the next tile's image is written
    } else {
      // position m = (32 wave + 8 rg) [uniform] + 4 lh [lane] + q: three disjoint bit ranges, so the element offset is
      // the sum of the parts' offsets -- one uniform 64-bit base per store, ONE per-lane byte offset (the epilogue's
      // address arithmetic, a 64-bit multiply chain per element, was ~400 of a tile's instructions; csrc/igemm3n.hip)
      constexpr uint32_t OSZ = ABF ? 2u : 4u;
      auto eoff = [&](int m) { return ((m >> P.lgTW) * p.W + (m & (p.TW - 1))) * p.out_pitch; };
      const int wu = __builtin_amdgcn_readfirstlane(wave);
      char* ob = reinterpret_cast<char*>(p.out) + (((size_t)nb * p.H + th0) * p.W + tw0) * p.out_pitch * OSZ;
      const uint32_t voff = (uint32_t)(eoff(4 * lh) + n) * OSZ;
      const bool full = th0 + p.TH <= p.H;
      auto put = [&](int rg, int q) {
        const int mu = wu * 32 + 8 * rg;
        const float v = acc[4 * rg + q] + bias;
        char* dst = ob + (size_t)(uint32_t)(eoff(mu) + eoff(q)) * OSZ + voff;
        if (ABF) *reinterpret_cast<__bf16*>(dst) = (__bf16)v;
        else *reinterpret_cast<float*>(dst) = v;
        if (STATS) { s0 += v; s1 = fmaf(v, v, s1); }
      };
      // the stores and the wait for the next tile's patch share an arm: the counted wait is only reached behind the
      // unmasked stores it counts (tools/asm_load_check.py checks the count on the listing)
      if (allfull) {   // uniform: sixteen unmasked stores, no branch
#pragma unroll
        for (int rg = 0; rg < 4; ++rg)
#pragma unroll
          for (int q = 0; q < 4; ++q) put(rg, q);
        patch_wait(true);
      } else {
#pragma unroll
        for (int rg = 0; rg < 4; ++rg)
#pragma unroll
          for (int q = 0; q < 4; ++q)
            if (nok && (full || th0 + ((wu * 32 + 8 * rg + 4 * lh + q) >> P.lgTW) < p.H)) put(rg, q);
        patch_wait(false);
      }
    }
  }
  if (STATS) {
    __syncthreads();
    float* red = reinterpret_cast<float*>(As);  // [4 waves][2][32]
    const float a = s0 + __shfl_xor(s0, 32, 64), b = s1 + __shfl_xor(s1, 32, 64);
    if (lh == 0) { red[(wave * 2 + 0) * 32 + li] = a; red[(wave * 2 + 1) * 32 + li] = b; }
    __syncthreads();
    if (tid < 64) {
      const int which = tid >> 5, c = tid & 31;
      if (n0 + c < p.N)
        p.stats[((size_t)blockIdx.x * 2 + which) * p.N + n0 + c] =
            red[(0 * 2 + which) * 32 + c] + red[(1 * 2 + which) * 32 + c] + red[(2 * 2 + which) * 32 + c] + red[(3 * 2 + which) * 32 + c];
    }
  }
}

template <int BN, int STATS, int RB, int PV>
static int launch_i3pv(const Igemm3Params& P, dim3 grid, size_t smem, hipStream_t s) {
  static BsedLdsOnce once;
  BSED_HIP(bsed_max_lds(once, (const void*)igemm3_kernel<BN, STATS, RB, PV>));
  hipLaunchKernelGGL((igemm3_kernel<BN, STATS, RB, PV>), grid, dim3(I3_THREADS), smem, s, P);
  BSED_LAUNCH_CHECK();
  return BSED_OK;
}

// PV = float4 patch elements per thread: ceil(PP * 8 / 256), instantiated for 6 (up to 192 patch positions: the 18x10
// patch of a 16x8 tile), 9 (288: tall narrow tiles) and 12 (384: the 256-position tiles)
template <int BN, int STATS, int RB>
static int launch_i3(const Igemm3Params& P, dim3 grid, size_t smem, hipStream_t s) {
  const int need = ceil_div(P.PP * 8, I3_THREADS);
  if (RB == 1 && need <= 6) return launch_i3pv<BN, STATS, RB, 6>(P, grid, smem, s);
  if (RB == 1 && need <= 9) return launch_i3pv<BN, STATS, RB, 9>(P, grid, smem, s);
  if (need <= 12) return launch_i3pv<BN, STATS, RB, 12>(P, grid, smem, s);
  bsed_set_error("bsed_igemm3: patch of %d positions exceeds the 384 supported", P.PP);
  return BSED_ERR_ARG;
}

extern "C" int bsed_igemm3(const BsedIgemmDesc* desc, void* stream) {
  BSED_CHECK_ARG(desc, "bsed_igemm3: null descriptor");
  Igemm3Params P;
  P.d = *desc;
  BsedIgemmDesc& d = P.d;
  BSED_CHECK_ARG(d.in && d.w && d.out, "bsed_igemm3: null tensor");
  BSED_CHECK_ARG(d.epilogue == BSED_EPI_PLAIN || d.epilogue == BSED_EPI_STATS, "bsed_igemm3: PLAIN / STATS epilogues only");
  BSED_CHECK_ARG(d.epilogue != BSED_EPI_STATS || d.stats, "bsed_igemm3: STATS needs a stats buffer");
  BSED_CHECK_ARG(d.NB > 0 && d.H > 0 && d.W > 0 && d.CIN > 0 && d.CIN % 32 == 0 && d.N > 0, "bsed_igemm3: CIN must be a multiple of 32");
  BSED_CHECK_ARG((d.TH * d.TW == I3_M || d.TH * d.TW == 2 * I3_M) && d.W % d.TW == 0,
                 "bsed_igemm3: TH*TW must be 128 or 256 (N a multiple of 128 only) and TW divide W");
  const int RB = d.TH * d.TW / I3_M;
  P.lgTW = 0;
  while ((1 << P.lgTW) < d.TW) ++P.lgTW;
  BSED_CHECK_ARG((1 << P.lgTW) == d.TW, "bsed_igemm3: TW must be a power of two");
  BSED_CHECK_ARG(d.ntaps >= 1 && d.ntaps <= 9, "bsed_igemm3: ntaps must be in 1..9");
  for (int t = 0; t < d.ntaps; ++t)
    BSED_CHECK_ARG(abs(d.dh[t]) <= d.hh && abs(d.dw[t]) <= d.hw, "bsed_igemm3: tap %d outside the halo", t);
  BSED_CHECK_ARG(d.in_pitch >= d.CIN && d.in_pitch % 4 == 0 && d.out_pitch >= d.N, "bsed_igemm3: bad pitch");
  BSED_CHECK_ARG(d.NP % 32 == 0 && d.NP >= d.N, "bsed_igemm3: NP must be N rounded up to 32");
  int BN = d.NP % 128 == 0 ? 128 : (d.NP % 64 == 0 ? 64 : 32);
  {  // A/B knob: BSED_IGEMM3_BN=64 runs the 128-channel layers as two 64-channel workgroups per tile
    static const int bn_cap = getenv("BSED_IGEMM3_BN") ? atoi(getenv("BSED_IGEMM3_BN")) : 128;
    if (RB == 1 && BN > bn_cap && (bn_cap == 64 || bn_cap == 32)) BN = bn_cap;
  }
  BSED_CHECK_ARG(RB == 1 || BN == 128, "bsed_igemm3: 256-position tiles are built for N a multiple of 128");
  d.tilesH = ceil_div(d.H, d.TH);
  d.tilesW = d.W / d.TW;
  P.PW = d.TW + 2 * d.hw;
  P.PH = d.TH + 2 * d.hh;
  P.PP = P.PW * P.PH;
  P.b_off = (P.PP * I3_ROW + 7) & ~7;
  P.pw_magic = ((1 << 20) + P.PW - 1) / P.PW;
  for (int pos = 0; pos < P.PP; ++pos)
    BSED_CHECK_ARG(((pos * P.pw_magic) >> 20) == pos / P.PW, "bsed_igemm3: internal: magic division fails for PW=%d", P.PW);
  size_t bytes = ((size_t)P.b_off + (size_t)BN * I3_ROW) * sizeof(unsigned short);
  bytes = std::max(bytes, (size_t)8 * BN * sizeof(float));
  BSED_CHECK_ARG(bytes <= 160 * 1024, "bsed_igemm3: tile needs %zu B of LDS", bytes);
  const long ntiles = (long)d.NB * d.tilesH * d.tilesW;
  BSED_CHECK_ARG(ntiles < (1L << 31), "bsed_igemm3: too many tiles");
  dim3 grid((unsigned)ntiles, d.NP / BN);
  hipStream_t s = (hipStream_t)stream;
  const bool st = d.epilogue == BSED_EPI_STATS;
  if (BN == 128 && RB == 2) return st ? launch_i3<128, 1, 2>(P, grid, bytes, s) : launch_i3<128, 0, 2>(P, grid, bytes, s);
  if (BN == 128) return st ? launch_i3<128, 1, 1>(P, grid, bytes, s) : launch_i3<128, 0, 1>(P, grid, bytes, s);
  if (BN == 64) return st ? launch_i3<64, 1, 1>(P, grid, bytes, s) : launch_i3<64, 0, 1>(P, grid, bytes, s);
  return st ? launch_i3<32, 1, 1>(P, grid, bytes, s) : launch_i3<32, 0, 1>(P, grid, bytes, s);
}

// ---- CIN = 16 variant: w = bsed_pack_weight3s table; persistent grid of G workgroups per 32 output channels;
// stats (STATS epilogue) has G rows (one per workgroup), not one per tile
extern "C" int bsed_pack_weight3s(const float* src, void* dst, int ntaps, int K, int N, int NP, long s_tap, long s_k,
                                  long s_n, void* stream) {
  BSED_CHECK_ARG(src && dst && ntaps > 0 && K > 0 && K % 16 == 0 && N > 0 && NP >= N && NP % 32 == 0,
                 "bsed_pack_weight3s: K must be a multiple of 16, NP a multiple of 32 >= N");
  const long total = (long)(NP / 32) * ntaps * (K / 16) * 64;
  hipLaunchKernelGGL(pack_weight3s_kernel, dim3((unsigned)ceil_div(total, 256)), dim3(256), 0, (hipStream_t)stream, src,
                     (unsigned short*)dst, ntaps, K / 16, N, NP, s_tap, s_k, s_n);
  BSED_LAUNCH_CHECK();
  return BSED_OK;
}

extern "C" int bsed_igemm3s_auto_g(void) {
  const char* v = getenv("BSED_IGEMM3S_G");   // A/B knob
  return v && atoi(v) > 0 ? atoi(v) : 1024;
}
// per shape: the 32 -> 16 channel data gradient (half-width weight table, 52 KB of LDS) runs three workgroups per CU:
// 768 / 1024 / 1536 workgroups 0.380 / 0.452 / 0.384 ms; the 16 -> 32 forward four: 0.406 / 0.365 / 0.409
extern "C" int bsed_igemm3s_auto_g2(int CIN, int N) {
  const char* v = getenv("BSED_IGEMM3S_G");
  if (v && atoi(v) > 0) return atoi(v);
  return (CIN == 32 && N <= 16) ? 768 : 1024;
}

extern "C" int bsed_igemm3s(const BsedIgemmDesc* desc, int G, void* stream) {
  BSED_CHECK_ARG(desc, "bsed_igemm3s: null descriptor");
  Igemm3Params P;
  P.d = *desc;
  BsedIgemmDesc& d = P.d;
  BSED_CHECK_ARG(d.in && d.w && d.out, "bsed_igemm3s: null tensor");
  BSED_CHECK_ARG(d.epilogue == BSED_EPI_PLAIN || d.epilogue == BSED_EPI_STATS, "bsed_igemm3s: PLAIN / STATS epilogues only");
  BSED_CHECK_ARG(d.epilogue != BSED_EPI_STATS || d.stats, "bsed_igemm3s: STATS needs a stats buffer");
  BSED_CHECK_ARG(d.NB > 0 && d.H > 0 && d.W > 0 && (d.CIN == 16 || d.CIN == 32) && d.N > 0,
                 "bsed_igemm3s: built for CIN = 16 or 32");
  const int KS = d.CIN / 16;
  BSED_CHECK_ARG(d.TH * d.TW == I3_M && d.W % d.TW == 0, "bsed_igemm3s: TH*TW must be 128 and TW divide W");
  P.lgTW = 0;
  while ((1 << P.lgTW) < d.TW) ++P.lgTW;
  BSED_CHECK_ARG((1 << P.lgTW) == d.TW, "bsed_igemm3s: TW must be a power of two");
  BSED_CHECK_ARG(d.ntaps == 9 || d.ntaps == 1, "bsed_igemm3s: 9 or 1 taps");
  for (int t = 0; t < d.ntaps; ++t)
    BSED_CHECK_ARG(abs(d.dh[t]) <= d.hh && abs(d.dw[t]) <= d.hw, "bsed_igemm3s: tap %d outside the halo", t);
  BSED_CHECK_ARG(d.in_pitch >= d.CIN && d.in_pitch % (d.act_bf16 ? 8 : 4) == 0 && d.out_pitch >= d.N, "bsed_igemm3s: bad pitch");
  BSED_CHECK_ARG(d.NP % 32 == 0 && d.NP >= d.N, "bsed_igemm3s: NP must be N rounded up to 32");
  d.tilesH = ceil_div(d.H, d.TH);
  d.tilesW = d.W / d.TW;
  P.PW = d.TW + 2 * d.hw;
  P.PH = d.TH + 2 * d.hh;
  P.PP = P.PW * P.PH;
  BSED_CHECK_ARG(P.PP <= (KS == 1 ? 256 : 192), "bsed_igemm3s: patch of %d positions exceeds the %d supported", P.PP,
                 KS == 1 ? 256 : 192);
  P.b_off = 0;
  P.pw_magic = ((1 << 20) + P.PW - 1) / P.PW;
  for (int pos = 0; pos < P.PP; ++pos)
    BSED_CHECK_ARG(((pos * P.pw_magic) >> 20) == pos / P.PW, "bsed_igemm3s: internal: magic division fails for PW=%d", P.PW);
  const long ntiles = (long)d.NB * d.tilesH * d.tilesW;
  BSED_CHECK_ARG(ntiles < (1L << 31) && G > 0 && G <= ntiles, "bsed_igemm3s: G must be in 1..%ld tiles", ntiles);
  // N <= 16: transposed epilogue (wave-private LDS image, float4 row stores)
  const bool nv16 = d.N <= 16 && d.N % 4 == 0 && d.out_pitch % 4 == 0;
  const size_t bytes = (size_t)d.ntaps * KS * 2 * (nv16 ? 32 : 64) * 16 + (size_t)P.PP * ((d.act_bf16 ? 1 : 2) * d.CIN + 8) * 2 +
                       (nv16 ? (size_t)4 * 32 * I3S_EROW * sizeof(float) : 0);
  dim3 grid((unsigned)G, d.NP / 32);
  hipStream_t s = (hipStream_t)stream;
  const bool st = d.epilogue == BSED_EPI_STATS;
#define I3S_LAUNCH1(S, T, K, V)                                                                                       \
  do {                                                                                                                \
    static BsedLdsOnce once, onceb;                                                                                   \
    if (d.act_bf16) {                                                                                                 \
      BSED_HIP(bsed_max_lds(onceb, (const void*)igemm3s_kernel<S, T, K, V, 1>));                                     \
      hipLaunchKernelGGL((igemm3s_kernel<S, T, K, V, 1>), grid, dim3(I3_THREADS), bytes, s, P);                       \
    } else {                                                                                                          \
      BSED_HIP(bsed_max_lds(once, (const void*)igemm3s_kernel<S, T, K, V, 0>));                                      \
      hipLaunchKernelGGL((igemm3s_kernel<S, T, K, V, 0>), grid, dim3(I3_THREADS), bytes, s, P);                       \
    }                                                                                                                 \
  } while (0)
#define I3S_LAUNCH(T, K)                                                                                              \
  do {                                                                                                                \
    if (st && nv16) I3S_LAUNCH1(1, T, K, 16);                                                                         \
    else if (st) I3S_LAUNCH1(1, T, K, 32);                                                                            \
    else if (nv16) I3S_LAUNCH1(0, T, K, 16);                                                                          \
    else I3S_LAUNCH1(0, T, K, 32);                                                                                    \
  } while (0)
  if (d.ntaps == 9 && KS == 1) I3S_LAUNCH(9, 1);
  else if (d.ntaps == 9) I3S_LAUNCH(9, 2);
  else if (KS == 1) I3S_LAUNCH(1, 1);
  else I3S_LAUNCH(1, 2);
#undef I3S_LAUNCH
#undef I3S_LAUNCH1
  BSED_LAUNCH_CHECK();
  return BSED_OK;
}

extern "C" int bsed_pack_weights_batch(const BsedPackJob* jobs, int njobs, void* stream) {
  BSED_CHECK_ARG(jobs && njobs > 0 && njobs <= BSED_PACK_MAX_JOBS, "bsed_pack_weights_batch: 1..%d jobs", BSED_PACK_MAX_JOBS);
  PkBatch Bt;
  Bt.njobs = njobs;
  long wg = 0;
  for (int i = 0; i < njobs; ++i) {
    const BsedPackJob& q = jobs[i];
    BSED_CHECK_ARG(q.src && q.dst && q.ntaps > 0 && q.K > 0 && q.N > 0 && q.NP >= q.N && q.NP % 32 == 0 &&
                   (q.kind == 0 ? q.K % 32 == 0 : (q.kind == 1 && q.K % 16 == 0)), "bsed_pack_weights_batch: bad job %d", i);
    PkJob& J = Bt.j[i];
    J.src = q.src; J.dst = (unsigned short*)q.dst; J.kind = q.kind; J.ntaps = q.ntaps; J.N = q.N; J.NP = q.NP;
    J.s_tap = q.s_tap; J.s_k = q.s_k; J.s_n = q.s_n;
    if (q.kind == 0) { J.KS = q.K / 32; J.total = (long)q.ntaps * J.KS * q.NP * 32; }
    else { J.KS = q.K / 16; J.total = (long)(q.NP / 32) * q.ntaps * J.KS * 64; }
    J.wg0 = (int)wg;
    wg += ceil_div(J.total, 256);
    BSED_CHECK_ARG(wg < (1L << 31), "bsed_pack_weights_batch: too many workgroups");
  }
  hipLaunchKernelGGL(pack_weights_batch_kernel, dim3((unsigned)wg), dim3(256), 0, (hipStream_t)stream, Bt);
  BSED_LAUNCH_CHECK();
  return BSED_OK;
}
