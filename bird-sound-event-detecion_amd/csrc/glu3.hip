// GLU stage (BatchNorm-apply -> Linear -> sigmoid gate -> Dropout -> AvgPool) in the split-fp32 ("bf16x3") contraction
// mode: forward and fused backward on v_mfma_f32_32x32x16_bf16, activations never staged through LDS.
//
// Reference math (src/models/CNN.py:5-16,59-67 and its autograd):
//   xn = y*scale + shift,  lin = xn W^T + b,  sig = sigmoid(xn),  res = lin*sig,  pooled = avgpool(dropout(res))
//   d_res = unpool(d_pooled) * mask/(1-p) / window
//   d_lin = d_res*sig                         -> db += sum d_lin,  dW += d_lin^T xn
//   g     = d_lin W + d_res*lin*sig*(1-sig)   = dL/d xn, plus the BatchNorm-backward sums (sum g, sum g*y)
//
// The fp32-core kernels (igemm.hip EPI_GLU_POOL, glu_bwd.hip) stage the activation tile in LDS and run at 25-70 TFLOP/s
// where the stage is HBM-bound by a wide margin on the bf16 cores.  Here every operand is fetched in the register
// layout its MFMA wants (lane maps: A[row = lane&31][k = 8*(lane>>5)+j], B[k = 8*(lane>>5)+j][col = lane&31],
// C[row = (r&3)+8*(r>>2)+4*(lane>>5)][col = lane&31]):
//   GEMM1 (lin = xn W^T): A fragments straight from global y (lane = position, 8 consecutive channels = 32 B),
//          BatchNorm applied and split into bf16 hi/lo in registers; B fragments (W^T, hi and lo) pre-split once per
//          workgroup into LDS in fragment order (16 B per lane, conflict-free).
//   epilogue 1: y is re-read (L1/L2-hot) in the C layout (lane = channel, 16 positions), d_lin and the gate term are
//          formed there; the gate term stays in the accumulators and seeds GEMM2.
//   GEMM2 (g = d_lin W + gate term): d_lin crosses from the C layout (lane = channel) to the A layout (lane =
//          position) through a wave-private bf16 hi|lo LDS tile -- the only activation-sized LDS traffic of the kernel.
//   GEMM3 (dW += d_lin^T xn, K = positions): the C layout IS the K-major operand layout: a lane holds 16 positions of
//          one channel, and both operands use the same position order, so d_lin (A) and xn (B) fragments are packed
//          from the registers that epilogue 1 already holds.  Each wave accumulates dW over its own 32 positions of
//          every tile (persistent accumulators) and writes one partial slab at the end (deterministic reduction by
//          bsed_reduce_partials; no float atomics).
// fp32 accumulation everywhere; the dropped lo*lo term is 2^-16 relative.
#include "bsed_common.h"
#include "../../include/bsed.h"
#include <algorithm>
#include <stdlib.h>

typedef short bf16x8 __attribute__((ext_vector_type(8)));
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));

#define G3_THREADS 256
#define G3_M 128

struct Glu3Params {
  const float* y; const float* scale; const float* shift;
  const float* w;      // (C,C) Linear weight, [n][c]
  const float* bias; const float* dpool;
  float* g; float* part_dw; float* part_db; float* part_st;
  float* pooled;       // forward output
  int NB, H, W, TH, TW, lgTW, tilesH, tilesW, ntiles;
  int ph, pw, Hp, Wp;
  float drop_p; uint32_t rng_stream; uint64_t seed;
  const uint64_t* seed_add;   // device-resident addend of the seed (HIP-graph replays), or null
};

static bool glu3_const_tw(const Glu3Params& P) {   // BSED_GLU3_LGTW=-1: runtime-geometry instances only (A/B runs)
  static const bool off = getenv("BSED_GLU3_LGTW") && atoi(getenv("BSED_GLU3_LGTW")) < 0;
  return P.lgTW == 4 && !off;
}

// Orders this wave's LDS writes before its later LDS reads.  The hardware executes one wave's LDS instructions in
// issue order, but the COMPILER may move a 16-byte fragment load above the 2-byte element stores that produce it (type
// based alias analysis sees unrelated types; __builtin_amdgcn_wave_barrier() does not order memory operations).
__device__ __forceinline__ void wave_lds_fence() { asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); }
// The same fence for the point where accumulators that VALU code has just written (the gate term) become the SrcC of
// the next MFMA.  Tying them to the asm makes the register allocator finish its copies into the accumulator tuples
// BEFORE the wait, and the s_nop pads the VALU-write -> MFMA-SrcC distance: without it hipcc (ROCm 7.2) left a
// `v_mov` into the tuple one `s_nop 0` ahead of the MFMA, and lanes 48..63 of that register (the last quarter a wave64
// VALU instruction writes) were occasionally read stale -- one wrong position per ~10 runs of a 2.5 M element tensor.
template <int NT>
__device__ __forceinline__ void acc_handoff_fence(f32x16 (&acc)[NT]) {
  if constexpr (NT == 1) asm volatile("s_waitcnt lgkmcnt(0)\n\ts_nop 7" : "+v"(acc[0]) :: "memory");
  else if constexpr (NT == 2) asm volatile("s_waitcnt lgkmcnt(0)\n\ts_nop 7" : "+v"(acc[0]), "+v"(acc[1]) :: "memory");
  else asm volatile("s_waitcnt lgkmcnt(0)\n\ts_nop 7" : "+v"(acc[0]), "+v"(acc[1]), "+v"(acc[2]), "+v"(acc[3]) :: "memory");
}
// one product of split operands: three MFMAs (lo*hi, hi*lo, hi*hi), or -- bf16 mode -- the hi*hi one alone
template <int ABF>
__device__ __forceinline__ f32x16 mfma_sp(const bf16x8& a_hi, const bf16x8& a_lo, const bf16x8& b_hi, const bf16x8& b_lo,
                                          f32x16 acc) {
  if (!ABF) {
    acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a_lo, b_hi, acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a_hi, b_lo, acc, 0, 0, 0);
  }
  return __builtin_amdgcn_mfma_f32_32x32x16_bf16(a_hi, b_hi, acc, 0, 0, 0);
}
__device__ __forceinline__ int crow3g(int r, int lh) { return (r & 3) + 8 * (r >> 2) + 4 * lh; }
__device__ __forceinline__ uint32_t rne_bits(float x) {  // bf16 round-to-nearest-even, result in the UPPER 16 bits
  const uint32_t u = __float_as_uint(x);
  return u + 0x7FFFu + ((u >> 16) & 1u);
}
// two fp32 values -> one packed word of bf16 hi parts and one of bf16 lo parts (element 0 in the low half)
__device__ __forceinline__ void split_pack2(float a, float b, uint32_t& hi, uint32_t& lo) { bsed_split2(a, b, hi, lo); }
__device__ __forceinline__ void split_pack8(const float* v, bf16x8& hi, bf16x8& lo) {
  u32x4 h, l;
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    uint32_t hh, ll;
    split_pack2(v[2 * i], v[2 * i + 1], hh, ll);
    h[i] = hh; l[i] = ll;
  }
  hi = __builtin_bit_cast(bf16x8, h);
  lo = __builtin_bit_cast(bf16x8, l);
}

// LDS fragment tables of the two weight operands: frag (j, ks, hi|lo) at ((j*KS + ks)*2 + hl)*64 + lane, 16 B each
//   WF (GEMM1 B = W^T): lane (n = 32j + li, lh) holds W[n][16ks + 8lh .. +8]
//   WB (GEMM2 B = W)  : lane (c = 32j + li, lh) holds W[16ks + 8lh .. +8][c]
template <int C>
__device__ __forceinline__ void build_weight_frags(const float* __restrict__ w, bf16x8* WF, bf16x8* WB, int tid) {
  constexpr int NT = C / 32, KS = C / 16;
  for (int f = tid; f < NT * KS * 64; f += G3_THREADS) {
    const int lane = f & 63, li = lane & 31, lh = lane >> 5;
    const int ks = (f >> 6) % KS, j = (f >> 6) / KS;
    float v[8];
    bf16x8 hi, lo;
#pragma unroll
    for (int q = 0; q < 8; ++q) v[q] = w[(size_t)(32 * j + li) * C + 16 * ks + 8 * lh + q];
    split_pack8(v, hi, lo);
    WF[((j * KS + ks) * 2 + 0) * 64 + lane] = hi;
    WF[((j * KS + ks) * 2 + 1) * 64 + lane] = lo;
    if (WB) {
#pragma unroll
      for (int q = 0; q < 8; ++q) v[q] = w[(size_t)(16 * ks + 8 * lh + q) * C + 32 * j + li];
      split_pack8(v, hi, lo);
      WB[((j * KS + ks) * 2 + 0) * 64 + lane] = hi;
      WB[((j * KS + ks) * 2 + 1) * 64 + lane] = lo;
    }
  }
}

// GEMM1 A fragments of this wave's 32 positions: BatchNorm-applied, split.  Two halves so that a kernel can request the
// NEXT tile's rows (a_frags_issue: raw values in registers) long before it needs them (a_frags_finish).
template <int C, int ABF>
__device__ __forceinline__ float a_frags_issue(const Glu3Params& P, int lgTW, int nb, int th0, int tw0, int wave, int li, int lh,
                                               float (&raw)[C / 16][8]) {
  constexpr int KS = C / 16;
  const int mA = wave * 32 + li;
  const int gh = th0 + (mA >> lgTW), gw = tw0 + (mA & ((1 << lgTW) - 1));
  const bool ok = gh < P.H;
  const size_t rp = (((size_t)nb * P.H + (ok ? gh : 0)) * P.W + gw) * C + 8 * lh;
#pragma unroll
  for (int ks = 0; ks < KS; ++ks) act_ld8<ABF>(P.y, rp + 16 * ks, raw[ks]);
  return ok ? 1.0f : 0.0f;
}
template <int C>
__device__ __forceinline__ void a_frags_finish(const float (&raw)[C / 16][8], float okf, const float* s_sc, const float* s_sh,
                                               int lh, bf16x8* a_hi, bf16x8* a_lo) {
  constexpr int KS = C / 16;
#pragma unroll
  for (int ks = 0; ks < KS; ++ks) {
    const float4 s0 = *reinterpret_cast<const float4*>(s_sc + 16 * ks + 8 * lh);
    const float4 s1 = *reinterpret_cast<const float4*>(s_sc + 16 * ks + 8 * lh + 4);
    const float4 h0 = *reinterpret_cast<const float4*>(s_sh + 16 * ks + 8 * lh);
    const float4 h1 = *reinterpret_cast<const float4*>(s_sh + 16 * ks + 8 * lh + 4);
    float v[8];
    v[0] = fmaf(raw[ks][0], s0.x, h0.x) * okf; v[1] = fmaf(raw[ks][1], s0.y, h0.y) * okf;
    v[2] = fmaf(raw[ks][2], s0.z, h0.z) * okf; v[3] = fmaf(raw[ks][3], s0.w, h0.w) * okf;
    v[4] = fmaf(raw[ks][4], s1.x, h1.x) * okf; v[5] = fmaf(raw[ks][5], s1.y, h1.y) * okf;
    v[6] = fmaf(raw[ks][6], s1.z, h1.z) * okf; v[7] = fmaf(raw[ks][7], s1.w, h1.w) * okf;
    split_pack8(v, a_hi[ks], a_lo[ks]);
  }
}
template <int C, int ABF>
__device__ __forceinline__ void load_a_frags(const Glu3Params& P, int lgTW, const float* s_sc, const float* s_sh, int nb,
                                             int th0, int tw0, int wave, int li, int lh, bf16x8* a_hi, bf16x8* a_lo) {
  float raw[C / 16][8];
  const float okf = a_frags_issue<C, ABF>(P, lgTW, nb, th0, tw0, wave, li, lh, raw);
  a_frags_finish<C>(raw, okf, s_sc, s_sh, lh, a_hi, a_lo);
}

__device__ float g3_sink[G3_M];

__device__ __forceinline__ float drop_mul32(uint32_t e, uint32_t key, uint32_t thr, float scale) {
  // == drop_mul(e, ...) of bsed_common.h for element indices below 2^32 (checked on the host)
  const uint32_t h = mix32(e * 0x9E3779B1u + key);
  return (h >> 8) >= thr ? scale : 0.f;
}

#ifndef G3_B32_WPE
#define G3_B32_WPE 3
#endif
template <int C, int LGTW, int ABF>
__global__ __launch_bounds__(G3_THREADS, C == 32 ? G3_B32_WPE : 2) void glu_bwd3_kernel(const Glu3Params P) {
  constexpr int NT = C / 32, KS = C / 16;
  // LGTW >= 0: the tile width is a compile-time constant (16 for every block of the reference network with more than
  // 8 frequency bins), so the per-row index arithmetic of the epilogues -- (m >> lgTW, m & (TW - 1)) of 16 rows per
  // lane, with their clamps, pooling indices and dropout counters: ~600 of the ~1400 instructions of a tile at
  // C = 32 -- folds into constants plus a lane term
  const int lgTW = LGTW >= 0 ? LGTW : P.lgTW;
  const int TWc = 1 << lgTW, THc = G3_M >> lgTW;
  constexpr int DP = 2 * C + 8;  // ushorts per row of the d_lin tile: C hi | C lo | 8 pad ((2C+8)*2 B = odd x 16 B)
  extern __shared__ __align__(16) unsigned char smem_raw[];
  bf16x8* WF = reinterpret_cast<bf16x8*>(smem_raw);
  bf16x8* WB = WF + NT * KS * 2 * 64;
  float* s_sc = reinterpret_cast<float*>(WB + NT * KS * 2 * 64);
  float* s_sh = s_sc + C;
  unsigned short* Dall = reinterpret_cast<unsigned short*>(s_sh + C);
  const int tid = threadIdx.x, wave = __builtin_amdgcn_readfirstlane(tid >> 6), lane = tid & 63, li = lane & 31, lh = lane >> 5;
  unsigned short* D = Dall + wave * 32 * DP;  // wave-private

  build_weight_frags<C>(P.w, WF, WB, tid);
  for (int i = tid; i < C; i += G3_THREADS) { s_sc[i] = P.scale[i]; s_sh[i] = P.shift[i]; }
  __syncthreads();

  const int sph = P.ph >> 1, spw = P.pw >> 1;
  const float inv_pool = 1.0f / (float)(P.ph * P.pw);
  const uint32_t dkey = drop_key(P.rng_stream, P.seed + (P.seed_add ? *P.seed_add : 0)), dthr = drop_threshold(P.drop_p);
  const float dscale = P.drop_p > 0.f ? 1.0f / (1.0f - P.drop_p) : 1.0f;

  float bias[NT], csc[NT], csh[NT], sdb[NT], sgs[NT], sgy[NT];
  f32x16 acc3[NT][NT];
#pragma unroll
  for (int j = 0; j < NT; ++j) {
    bias[j] = P.bias[32 * j + li];
    csc[j] = s_sc[32 * j + li]; csh[j] = s_sh[32 * j + li];
    sdb[j] = 0.f; sgs[j] = 0.f; sgy[j] = 0.f;
#pragma unroll
    for (int jc = 0; jc < NT; ++jc)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc3[j][jc][r] = 0.f;
  }

  // C = 32 (three waves per SIMD with registers to spare): the next tile's A rows are requested right after this tile's
  // GEMM1 -- one of the tile's three dependent memory waits gone, and the lines are in L2 when epilogue 1 re-reads them
  constexpr bool PREF = C == 32 && LGTW >= 0;   // (the runtime-width build would spill)
  constexpr bool KEEPY = C == 32 && LGTW >= 0;
  constexpr bool G2 = C == 32 && LGTW >= 0;     // epilogue 1 requests two row groups (16 loads) at a time
  float nraw[KS][8];
  float nokf = 0.f;
  auto tile_origin = [&](int t, int& nb_, int& th0_, int& tw0_) {
    const int tw_i = t % P.tilesW; t /= P.tilesW;
    const int th_i = t % P.tilesH;
    nb_ = t / P.tilesH; th0_ = th_i * THc; tw0_ = tw_i * TWc;
  };
  if (PREF && (int)blockIdx.x < P.ntiles) {
    int nb_, th0_, tw0_;
    tile_origin(blockIdx.x, nb_, th0_, tw0_);
    nokf = a_frags_issue<C, ABF>(P, lgTW, nb_, th0_, tw0_, wave, li, lh, nraw);
  }
  for (int tile0 = blockIdx.x; tile0 < P.ntiles; tile0 += gridDim.x) {
    int nb, th0, tw0;
    tile_origin(tile0, nb, th0, tw0);
    // opaque copy of the lane's row offset: keeps the per-row index arithmetic of the epilogues (16 rows x several
    // values, all tile-invariant) from being hoisted out of the tile loop into ~40 long-lived registers
    int lhv = 4 * lh;
    if (LGTW < 0) asm volatile("" : "+v"(lhv));

    // ---- GEMM1: lin = xn W^T
    f32x16 acc[NT];
    {
      bf16x8 a_hi[KS], a_lo[KS];
      if (PREF) a_frags_finish<C>(nraw, nokf, s_sc, s_sh, lh, a_hi, a_lo);
      else load_a_frags<C, ABF>(P, lgTW, s_sc, s_sh, nb, th0, tw0, wave, li, lh, a_hi, a_lo);
#pragma unroll
      for (int j = 0; j < NT; ++j)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[j][r] = 0.f;
#pragma unroll
      for (int ks = 0; ks < KS; ++ks)
#pragma unroll
        for (int j = 0; j < NT; ++j) {
          const bf16x8 b_hi = WF[((j * KS + ks) * 2 + 0) * 64 + lane];
          const bf16x8 b_lo = WF[((j * KS + ks) * 2 + 1) * 64 + lane];
          acc[j] = mfma_sp<ABF>(a_hi[ks], a_lo[ks], b_hi, b_lo, acc[j]);
        }
    }
    if (PREF) {   // (the last tile re-requests its own rows: no branch around the loads)
      const int nt = tile0 + (int)gridDim.x < P.ntiles ? tile0 + (int)gridDim.x : tile0;
      int nb_, th0_, tw0_;
      tile_origin(nt, nb_, th0_, tw0_);
      nokf = a_frags_issue<C, ABF>(P, lgTW, nb_, th0_, tw0_, wave, li, lh, nraw);
    }

    // ---- epilogue 1 (C layout: lane = channel 32j+li, register r = position wave*32 + crow(r, lh)).
    // d_lin and xn leave it already packed as the GEMM3 operand fragments (frag f = registers 8f..8f+7).
    bf16x8 da_hi[NT][2], da_lo[NT][2], xb_hi[NT][2], xb_lo[NT][2];
    // C = 32: the 16 y values of the lane stay in registers for epilogue 2 (sum g y) instead of being read again
    uint32_t ykeep[KEEPY ? 16 : 1][NT];
#pragma unroll
    for (int f = 0; f < 2; ++f) {
      float dl8[NT][8], xn8[NT][8];
      // branch-free: rows below the image / outside the pooled extent read a clamped address and are masked.
      // The loads of a group (G2: of both row groups of the fragment) are ALL issued before the first conversion: left
      // alone, hipcc puts each pair of loads directly in front of its use -- one L2 round trip per row, serialised
      // (C = 32 backward 0.78 -> 0.68 ms with the fence alone).
      uint32_t dvr[2][4][NT], yvr[2][4][NT];
      float mk[2][4];
      uint32_t posv[2][4];
      auto issue = [&](int h) {
        const int rg = 2 * f + h;
#pragma unroll
        for (int rr = 0; rr < 4; ++rr) {
          const int mm = wave * 32 + 8 * rg + lhv + rr;
          const int gh = th0 + (mm >> lgTW), gw = tw0 + (mm & (TWc - 1));
          const int gph = gh >> sph, gpw = gw >> spw;
          mk[h][rr] = (gh < P.H && gph < P.Hp && gpw < P.Wp) ? inv_pool : 0.f;
          posv[h][rr] = ((uint32_t)(nb * P.H + min(gh, P.H - 1)) * (uint32_t)P.W + gw) * C + li;
          const uint32_t dpo = ((uint32_t)(nb * P.Hp + min(gph, P.Hp - 1)) * (uint32_t)P.Wp + min(gpw, P.Wp - 1)) * C + li;
#pragma unroll
          for (int j = 0; j < NT; ++j) {
            dvr[h][rr][j] = act_ld_raw<ABF>(P.dpool, dpo + 32 * j);
            yvr[h][rr][j] = act_ld_raw<ABF>(P.y, posv[h][rr] + 32 * j);
          }
        }
      };
      if (G2) { issue(0); issue(1); __builtin_amdgcn_sched_barrier(0); }
#pragma unroll
      for (int h = 0; h < 2; ++h) {
        const int rg = 2 * f + h;
        if (!G2) { issue(h); __builtin_amdgcn_sched_barrier(0); }
#pragma unroll
        for (int rr = 0; rr < 4; ++rr) {
          const int r = 4 * rg + rr;
          const int row = 8 * rg + lhv + rr;
#pragma unroll
          for (int j = 0; j < NT; ++j) {
            const int n = 32 * j + li;
            if (KEEPY) ykeep[r][j] = yvr[h][rr][j];
            const float xn = fmaf(act_cvt<ABF>(yvr[h][rr][j]), csc[j], csh[j]);
            const float sg = sigmoid_fast(xn);
            const float lin = acc[j][r] + bias[j];
            const float dres = act_cvt<ABF>(dvr[h][rr][j]) * mk[h][rr] * drop_mul32(posv[h][rr] + 32 * j, dkey, dthr, dscale);
            const float dl = dres * sg;
            const float tt = dres * lin * sg * (1.0f - sg);
            sdb[j] += dl;
            dl8[j][4 * h + rr] = dl;
            xn8[j][4 * h + rr] = xn;
            acc[j][r] = tt;
            const uint32_t hb = rne_bits(dl) & 0xFFFF0000u;
            D[row * DP + n] = (unsigned short)(hb >> 16);
            D[row * DP + C + n] = (unsigned short)(rne_bits(dl - __uint_as_float(hb)) >> 16);
          }
        }
      }
#pragma unroll
      for (int j = 0; j < NT; ++j) {
        split_pack8(dl8[j], da_hi[j][f], da_lo[j][f]);
        split_pack8(xn8[j], xb_hi[j][f], xb_lo[j][f]);
      }
    }
    acc_handoff_fence<NT>(acc);  // the tile is wave-private: LDS operations of one wave execute in issue order

    // ---- GEMM2: g = d_lin W + gate term (already in acc)
#pragma unroll
    for (int ks = 0; ks < KS; ++ks) {
      const bf16x8 d_hi = *reinterpret_cast<const bf16x8*>(D + li * DP + 16 * ks + 8 * lh);
      const bf16x8 d_lo = *reinterpret_cast<const bf16x8*>(D + li * DP + C + 16 * ks + 8 * lh);
#pragma unroll
      for (int j = 0; j < NT; ++j) {
        const bf16x8 b_hi = WB[((j * KS + ks) * 2 + 0) * 64 + lane];
        const bf16x8 b_lo = WB[((j * KS + ks) * 2 + 1) * 64 + lane];
        acc[j] = mfma_sp<ABF>(d_hi, d_lo, b_hi, b_lo, acc[j]);
      }
    }
    wave_lds_fence();

    // ---- GEMM3: dW[n][c] += sum over this wave's 32 positions of d_lin[p][n] xn[p][c]; operands from registers
#pragma unroll
    for (int jc = 0; jc < NT; ++jc)
#pragma unroll
      for (int jn = 0; jn < NT; ++jn)
#pragma unroll
        for (int f = 0; f < 2; ++f) {
          acc3[jn][jc] = mfma_sp<ABF>(da_hi[jn][f], da_lo[jn][f], xb_hi[jc][f], xb_lo[jc][f], acc3[jn][jc]);
        }

    // ---- epilogue 2: write g, BatchNorm-backward sums (y re-read: cache-hot)
#pragma unroll
    for (int rg = 0; rg < 4; ++rg) {
      uint32_t yvr[4][NT];
      float okf[4];
      uint32_t posv[4];
      float* gdst[4];
      uint32_t gidx[4];
#pragma unroll
      for (int rr = 0; rr < 4; ++rr) {
        const int mm = wave * 32 + 8 * rg + lhv + rr;
        const int gh = th0 + (mm >> lgTW), gw = tw0 + (mm & (TWc - 1));
        okf[rr] = gh < P.H ? 1.0f : 0.0f;
        posv[rr] = ((uint32_t)(nb * P.H + min(gh, P.H - 1)) * (uint32_t)P.W + gw) * C + li;
        gdst[rr] = gh < P.H ? P.g : g3_sink;  // rows below the image store to a sink: no branch
        gidx[rr] = gh < P.H ? posv[rr] : (uint32_t)li;
#pragma unroll
        for (int j = 0; j < NT; ++j) yvr[rr][j] = KEEPY ? ykeep[4 * rg + rr][j] : act_ld_raw<ABF>(P.y, posv[rr] + 32 * j);
      }
      if (!KEEPY) __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int rr = 0; rr < 4; ++rr) {
#pragma unroll
        for (int j = 0; j < NT; ++j) {
          const float gv = acc[j][4 * rg + rr] * okf[rr];  // (rows below the image carry g = 0 anyway: d_lin = gate = 0)
          act_st<ABF>(gdst[rr], gidx[rr] + 32 * j, gv);
          sgs[j] += gv;
          sgy[j] = fmaf(gv, act_cvt<ABF>(yvr[rr][j]), sgy[j]);
        }
      }
    }
  }

  // ---- partials: one dW slab per WAVE (4 per workgroup), db / BN sums per workgroup
#pragma unroll
  for (int jn = 0; jn < NT; ++jn)
#pragma unroll
    for (int jc = 0; jc < NT; ++jc)
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int n = 32 * jn + crow3g(r, lh);
        P.part_dw[(((size_t)blockIdx.x * 4 + wave) * C + n) * C + 32 * jc + li] = acc3[jn][jc][r];
      }
  __syncthreads();
  float* red = reinterpret_cast<float*>(Dall);  // [4 waves][3][C]
#pragma unroll
  for (int j = 0; j < NT; ++j) {
    const float a = sdb[j] + __shfl_xor(sdb[j], 32, 64);
    const float b = sgs[j] + __shfl_xor(sgs[j], 32, 64);
    const float c = sgy[j] + __shfl_xor(sgy[j], 32, 64);
    if (lh == 0) {
      red[(wave * 3 + 0) * C + 32 * j + li] = a;
      red[(wave * 3 + 1) * C + 32 * j + li] = b;
      red[(wave * 3 + 2) * C + 32 * j + li] = c;
    }
  }
  __syncthreads();
  for (int e = tid; e < 3 * C; e += G3_THREADS) {
    const int which = e / C, n = e % C;
    const float s = red[(0 * 3 + which) * C + n] + red[(1 * 3 + which) * C + n] + red[(2 * 3 + which) * C + n] +
                    red[(3 * 3 + which) * C + n];
    if (which == 0) {
      P.part_db[((size_t)blockIdx.x * 2 + 0) * C + n] = s;
      P.part_db[((size_t)blockIdx.x * 2 + 1) * C + n] = 0.f;
    } else {
      P.part_st[((size_t)blockIdx.x * 2 + (which - 1)) * C + n] = s;
    }
  }
}

// ---------------------------------------------------------------------------------------------------------------
// C = 128: the same backward WITHOUT the in-kernel weight gradient.  128 x 128 accumulators per wave do not fit in
// registers, so d_lin is also written to HBM and dW = d_lin^T xn runs as a 1-tap bsed_wgrad3 afterwards (BatchNorm
// applied on load there).  Both weight operands (hi|lo: 128 KB) sit in LDS, copied from a table that
// glu3_pack_frags_kernel builds once per call; d_lin crosses to the A layout of GEMM2 through a wave-private tile
// that holds ONE 32-channel quarter at a time (GEMM2's K loop is split in four), which keeps LDS at 148 KB.
__global__ void glu3_pack_frags_kernel(const float* __restrict__ w, bf16x8* __restrict__ table) {
  constexpr int C = 128, NT = 4, KS = 8;
  const int f = blockIdx.x * blockDim.x + threadIdx.x;
  if (f >= NT * KS * 64) return;
  const int lane = f & 63, li = lane & 31, lh = lane >> 5;
  const int ks = (f >> 6) % KS, j = (f >> 6) / KS;
  bf16x8* WF = table;
  bf16x8* WB = table + NT * KS * 2 * 64;
  float v[8];
  bf16x8 hi, lo;
#pragma unroll
  for (int q = 0; q < 8; ++q) v[q] = w[(size_t)(32 * j + li) * C + 16 * ks + 8 * lh + q];
  split_pack8(v, hi, lo);
  WF[((j * KS + ks) * 2 + 0) * 64 + lane] = hi;
  WF[((j * KS + ks) * 2 + 1) * 64 + lane] = lo;
#pragma unroll
  for (int q = 0; q < 8; ++q) v[q] = w[(size_t)(16 * ks + 8 * lh + q) * C + 32 * j + li];
  split_pack8(v, hi, lo);
  WB[((j * KS + ks) * 2 + 0) * 64 + lane] = hi;
  WB[((j * KS + ks) * 2 + 1) * 64 + lane] = lo;
}

// Eight waves share the 128 KB of weights: waves 0-3 and 4-7 work on two different 128-position tiles (no barrier
// inside the tile loop), which puts two waves on every SIMD instead of one.
#define G3N_THREADS 512
template <int ABF>
__global__ __launch_bounds__(G3N_THREADS) void glu_bwd3n_kernel(const Glu3Params P, const bf16x8* __restrict__ table,
                                                                float* __restrict__ dlin) {
  constexpr int C = 128, NT = 4, KS = 8;
#ifdef G3N_LGTW   // diagnostic builds (tools/build_variant.sh): tile geometry as a compile-time constant
  constexpr int g3n_lgTW = G3N_LGTW, g3n_TW = 1 << G3N_LGTW;
#else
  const int g3n_lgTW = P.lgTW, g3n_TW = P.TW;
#endif
  // Tile geometry is read from the parameter block (-DG3N_LGTW=<n>: compile-time, diagnostic).  Round 2 found builds
  // with compile-time geometry NOT bitwise repeatable (whole elements of g / d_lin changed between launches with every
  // tolerance test green).  Round 3 reproduced it (G3N_LGTW=4, 216 x 16 map: EVERY launch differs -- the pool / dropout
  // mask of the first row group flips for lanes 48..63 of channel tiles 1..3) and ruled out, by assembling patched
  // listings (tools/asm_variant.sh), every wait-state explanation: an s_nop after each vector instruction of the
  // epilogue, after each v_cmp, before each s_and_b64, after each load, s_waitcnt 0 everywhere -- none changes it.  It
  // is a code-generation problem of hipcc 7.2 tied to the SLP vectorizer's packed (v_pk_* with op_sel swizzles) form
  // of this epilogue: with -fno-slp-vectorize (csrc/build.sh, this file) BOTH geometries are bit-repeatable at all
  // four tile widths over 100 launches each (tools/glu3n_repeat.py), and the kernel is 5-12 % faster.  Compile-time
  // geometry is slower here (hoisted index arithmetic spills: 533 vs 423 us at 216 x 16), so it stays a runtime value.
  constexpr int DQ = 2 * 16 + 8;  // ushorts per row of the 16-channel chunk tile: 16 hi | 16 lo | 8 pad (80 B = 5 x 16 B)
  extern __shared__ __align__(16) unsigned char smem_raw[];
  bf16x8* WF = reinterpret_cast<bf16x8*>(smem_raw);
  bf16x8* WB = WF + NT * KS * 2 * 64;
  float* s_sc = reinterpret_cast<float*>(WB + NT * KS * 2 * 64);
  float* s_sh = s_sc + C;
  unsigned short* Dall = reinterpret_cast<unsigned short*>(s_sh + C);
  const int tid = threadIdx.x, wave8 = __builtin_amdgcn_readfirstlane(tid >> 6), lane = tid & 63, li = lane & 31, lh = lane >> 5;
  const int wave = wave8 & 3, half = wave8 >> 2;  // wave = 32-row block of the tile, half = which of the two tiles
  unsigned short* D = Dall + wave8 * 32 * DQ;  // wave-private

  for (int i = tid; i < 2 * NT * KS * 2 * 64; i += G3N_THREADS) WF[i] = table[i];
  for (int i = tid; i < C; i += G3N_THREADS) { s_sc[i] = P.scale[i]; s_sh[i] = P.shift[i]; }
  __syncthreads();

  const int sph = P.ph >> 1, spw = P.pw >> 1;
  const float inv_pool = 1.0f / (float)(P.ph * P.pw);
  const uint32_t dkey = drop_key(P.rng_stream, P.seed + (P.seed_add ? *P.seed_add : 0)), dthr = drop_threshold(P.drop_p);
  const float dscale = P.drop_p > 0.f ? 1.0f / (1.0f - P.drop_p) : 1.0f;

  float bias[NT], csc[NT], csh[NT], sdb[NT], sgs[NT], sgy[NT];
#pragma unroll
  for (int j = 0; j < NT; ++j) {
    bias[j] = P.bias[32 * j + li];
    csc[j] = s_sc[32 * j + li]; csh[j] = s_sh[32 * j + li];
    sdb[j] = 0.f; sgs[j] = 0.f; sgy[j] = 0.f;
  }

  for (int tile0 = 2 * blockIdx.x + half; tile0 < P.ntiles; tile0 += 2 * gridDim.x) {
    int tile = tile0;
    const int tw_i = tile % P.tilesW; tile /= P.tilesW;
    const int th_i = tile % P.tilesH;
    const int nb = tile / P.tilesH;
    const int th0 = th_i * P.TH, tw0 = tw_i * P.TW;
    int lhv = 4 * lh;
    asm volatile("" : "+v"(lhv));

    // ---- GEMM1: lin = xn W^T
    f32x16 acc[NT];
    {
      bf16x8 a_hi[KS], a_lo[KS];
      load_a_frags<C, ABF>(P, g3n_lgTW, s_sc, s_sh, nb, th0, tw0, wave, li, lh, a_hi, a_lo);
#pragma unroll
      for (int j = 0; j < NT; ++j)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[j][r] = 0.f;
#pragma unroll
      for (int ks = 0; ks < KS; ++ks)
#pragma unroll
        for (int j = 0; j < NT; ++j) {
          const bf16x8 b_hi = WF[((j * KS + ks) * 2 + 0) * 64 + lane];
          const bf16x8 b_lo = WF[((j * KS + ks) * 2 + 1) * 64 + lane];
          acc[j] = mfma_sp<ABF>(a_hi[ks], a_lo[ks], b_hi, b_lo, acc[j]);
        }
    }

    // ---- epilogue 1: gate term -> accumulators, d_lin -> HBM (for the weight-gradient pass) and, split into bf16
    // hi/lo pairs, kept in registers until its 32-channel quarter of GEMM2 comes up
    uint32_t dph[NT][8], dpl[NT][8];  // packed (r even | r odd << 16)
#pragma unroll
    for (int rg = 0; rg < 4; ++rg) {
      uint32_t dvr[4][NT], yvr[4][NT];
      float mk[4];
      uint32_t posv[4];
      float* dlrow[4];
      uint32_t dlidx[4];
#pragma unroll
      for (int rr = 0; rr < 4; ++rr) {
        const int mm = wave * 32 + 8 * rg + lhv + rr;
        const int gh = th0 + (mm >> g3n_lgTW), gw = tw0 + (mm & (g3n_TW - 1));
        const int gph = gh >> sph, gpw = gw >> spw;
        mk[rr] = (gh < P.H && gph < P.Hp && gpw < P.Wp) ? inv_pool : 0.f;
        posv[rr] = ((uint32_t)(nb * P.H + min(gh, P.H - 1)) * (uint32_t)P.W + gw) * C + li;
        dlrow[rr] = gh < P.H ? dlin : g3_sink;
        dlidx[rr] = gh < P.H ? posv[rr] : (uint32_t)li;
        const uint32_t dpo = ((uint32_t)(nb * P.Hp + min(gph, P.Hp - 1)) * (uint32_t)P.Wp + min(gpw, P.Wp - 1)) * C + li;
#pragma unroll
        for (int j = 0; j < NT; ++j) {
          dvr[rr][j] = act_ld_raw<ABF>(P.dpool, dpo + 32 * j);
          yvr[rr][j] = act_ld_raw<ABF>(P.y, posv[rr] + 32 * j);
        }
      }
      __builtin_amdgcn_sched_barrier(0);   // the group's loads are all issued before the first conversion
#pragma unroll
      for (int rp = 0; rp < 2; ++rp)
#pragma unroll
        for (int j = 0; j < NT; ++j) {
          float dl2[2];
#pragma unroll
          for (int e = 0; e < 2; ++e) {
            const int rr = 2 * rp + e, r = 4 * rg + rr;
            const float xn = fmaf(act_cvt<ABF>(yvr[rr][j]), csc[j], csh[j]);
            const float sg = sigmoid_fast(xn);
            const float lin = acc[j][r] + bias[j];
            const float dres = act_cvt<ABF>(dvr[rr][j]) * mk[rr] * drop_mul32(posv[rr] + 32 * j, dkey, dthr, dscale);
            const float dl = dres * sg;
            acc[j][r] = dres * lin * sg * (1.0f - sg);
            sdb[j] += dl;
            act_st<ABF>(dlrow[rr], dlidx[rr] + 32 * j, dl);
            dl2[e] = dl;
          }
          split_pack2(dl2[0], dl2[1], dph[j][2 * rg + rp], dpl[j][2 * rg + rp]);
        }
    }

    // ---- GEMM2 in eight 16-channel K chunks: g = d_lin W + gate term (already in acc)
    acc_handoff_fence<NT>(acc);
#pragma unroll
    for (int q = 0; q < 8; ++q) {
      // chunk q = channels 16q..16q+15 = C-layout tile j = q/2, lanes li in [16*(q&1), 16*(q&1) + 16)
      if ((li >> 4) == (q & 1)) {
#pragma unroll
        for (int i = 0; i < 8; ++i) {
          const int r = 2 * i;  // packed pair (r, r+1): rows crow(r), crow(r)+1
          const int row = 8 * (r >> 2) + lhv + (r & 3);
          D[row * DQ + (li & 15)] = (unsigned short)(dph[q >> 1][i] & 0xFFFFu);
          D[(row + 1) * DQ + (li & 15)] = (unsigned short)(dph[q >> 1][i] >> 16);
          D[row * DQ + 16 + (li & 15)] = (unsigned short)(dpl[q >> 1][i] & 0xFFFFu);
          D[(row + 1) * DQ + 16 + (li & 15)] = (unsigned short)(dpl[q >> 1][i] >> 16);
        }
      }
      wave_lds_fence();
      const bf16x8 d_hi = *reinterpret_cast<const bf16x8*>(D + li * DQ + 8 * lh);
      const bf16x8 d_lo = *reinterpret_cast<const bf16x8*>(D + li * DQ + 16 + 8 * lh);
#pragma unroll
      for (int j = 0; j < NT; ++j) {
        const bf16x8 b_hi = WB[((j * KS + q) * 2 + 0) * 64 + lane];
        const bf16x8 b_lo = WB[((j * KS + q) * 2 + 1) * 64 + lane];
        acc[j] = mfma_sp<ABF>(d_hi, d_lo, b_hi, b_lo, acc[j]);
      }
      wave_lds_fence();
    }

    // ---- epilogue 2: write g, BatchNorm-backward sums (y re-read: cache-hot)
#pragma unroll
    for (int rg = 0; rg < 4; ++rg) {
      uint32_t yvr[4][NT];
      float okf[4];
      uint32_t posv[4];
      float* gdst[4];
      uint32_t gidx[4];
#pragma unroll
      for (int rr = 0; rr < 4; ++rr) {
        const int mm = wave * 32 + 8 * rg + lhv + rr;
        const int gh = th0 + (mm >> g3n_lgTW), gw = tw0 + (mm & (g3n_TW - 1));
        okf[rr] = gh < P.H ? 1.0f : 0.0f;
        posv[rr] = ((uint32_t)(nb * P.H + min(gh, P.H - 1)) * (uint32_t)P.W + gw) * C + li;
        gdst[rr] = gh < P.H ? P.g : g3_sink;
        gidx[rr] = gh < P.H ? posv[rr] : (uint32_t)li;
#pragma unroll
        for (int j = 0; j < NT; ++j) yvr[rr][j] = act_ld_raw<ABF>(P.y, posv[rr] + 32 * j);
      }
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int rr = 0; rr < 4; ++rr)
#pragma unroll
        for (int j = 0; j < NT; ++j) {
          const float gv = acc[j][4 * rg + rr] * okf[rr];
          act_st<ABF>(gdst[rr], gidx[rr] + 32 * j, gv);
          sgs[j] += gv;
          sgy[j] = fmaf(gv, act_cvt<ABF>(yvr[rr][j]), sgy[j]);
        }
    }
  }

  __syncthreads();
  float* red = reinterpret_cast<float*>(Dall);  // [8 waves][3][C] = 12 KB of the 20 KB tile area
#pragma unroll
  for (int j = 0; j < NT; ++j) {
    const float a = sdb[j] + __shfl_xor(sdb[j], 32, 64);
    const float b = sgs[j] + __shfl_xor(sgs[j], 32, 64);
    const float c = sgy[j] + __shfl_xor(sgy[j], 32, 64);
    if (lh == 0) {
      red[(wave8 * 3 + 0) * C + 32 * j + li] = a;
      red[(wave8 * 3 + 1) * C + 32 * j + li] = b;
      red[(wave8 * 3 + 2) * C + 32 * j + li] = c;
    }
  }
  __syncthreads();
  for (int e = tid; e < 3 * C; e += G3N_THREADS) {
    const int which = e / C, n = e % C;
    float s = 0.f;
#pragma unroll
    for (int wv = 0; wv < 8; ++wv) s += red[(wv * 3 + which) * C + n];
    if (which == 0) {
      P.part_db[((size_t)blockIdx.x * 2 + 0) * C + n] = s;
      P.part_db[((size_t)blockIdx.x * 2 + 1) * C + n] = 0.f;
    } else {
      P.part_st[((size_t)blockIdx.x * 2 + (which - 1)) * C + n] = s;
    }
  }
}

// ---------------------------------------------------------------------------------------------------------------
// Forward: y -> BatchNorm-apply -> Linear -> sigmoid gate -> Dropout -> AvgPool -> pooled, one pass over y.
// GEMM1 as above; the epilogue works in the C layout where a lane holds, for its channel, four consecutive positions
// per register group: the (1,2) and (2,2) pooling windows of a tile row are lane-local (the vertical partner of a
// position sits 8 registers further at TW = 16, 4 at TW = 8, 2 at TW = 2), so pooling needs neither LDS nor shuffles.
template <int C, int LGTW, int ABF>
__global__ __launch_bounds__(G3_THREADS, 2) void glu_fwd3_kernel(const Glu3Params P) {
  constexpr int NT = C / 32, KS = C / 16;
  // LGTW >= 0: the tile width is a compile-time constant (16 for every block of the reference network with more than
  // 8 frequency bins), so the per-row index arithmetic of the epilogues -- (m >> lgTW, m & (TW - 1)) of 16 rows per
  // lane, with their clamps, pooling indices and dropout counters: ~600 of the ~1400 instructions of a tile at
  // C = 32 -- folds into constants plus a lane term
  const int lgTW = LGTW >= 0 ? LGTW : P.lgTW;
  const int TWc = 1 << lgTW, THc = G3_M >> lgTW;
  extern __shared__ __align__(16) unsigned char smem_raw[];
  bf16x8* WF = reinterpret_cast<bf16x8*>(smem_raw);
  float* s_sc = reinterpret_cast<float*>(WF + NT * KS * 2 * 64);
  float* s_sh = s_sc + C;
  const int tid = threadIdx.x, wave = __builtin_amdgcn_readfirstlane(tid >> 6), lane = tid & 63, li = lane & 31, lh = lane >> 5;
  // C <= 64: the wave's 32 position rows of y are staged once, as whole lines (float4 per lane, 1 KB of consecutive
  // addresses per load instruction), into a wave-private LDS image that serves both reads of y -- the A fragments
  // (8 consecutive channels of the lane's position: 16-byte pieces of 32 different lines per instruction when read
  // straight from global) and the gate input of the epilogue.  No workgroup barrier: the image is private to the wave.
  constexpr bool STAGE = C <= 64;
  constexpr int YROW = C + 4;  // floats per staged row: (C + 4) * 4 bytes = odd multiple of 16
  float* Ys = s_sh + C + wave * 32 * YROW;

  build_weight_frags<C>(P.w, WF, nullptr, tid);
  for (int i = tid; i < C; i += G3_THREADS) { s_sc[i] = P.scale[i]; s_sh[i] = P.shift[i]; }
  __syncthreads();

  const int sph = P.ph >> 1, spw = P.pw >> 1;
  const float inv_pool = 1.0f / (float)(P.ph * P.pw);
  const uint32_t dkey = drop_key(P.rng_stream, P.seed + (P.seed_add ? *P.seed_add : 0)), dthr = drop_threshold(P.drop_p);
  const float dscale = P.drop_p > 0.f ? 1.0f / (1.0f - P.drop_p) : 1.0f;
  // register distance of the vertical pooling partner (position m + TW): crow(r + dr) = crow(r) + TW
  const int TW = TWc;
  float bias[NT], csc[NT], csh[NT];
#pragma unroll
  for (int j = 0; j < NT; ++j) {
    bias[j] = P.bias[32 * j + li];
    csc[j] = s_sc[32 * j + li]; csh[j] = s_sh[32 * j + li];
  }

  for (int tile0 = blockIdx.x; tile0 < P.ntiles; tile0 += gridDim.x) {
    int tile = tile0;
    const int tw_i = tile % P.tilesW; tile /= P.tilesW;
    const int th_i = tile % P.tilesH;
    const int nb = tile / P.tilesH;
    const int th0 = th_i * THc, tw0 = tw_i * TWc;
    int lhv = 4 * lh;
    if (C > 64) asm volatile("" : "+v"(lhv));  // C <= 64 have registers to spare: let the per-row index math be hoisted

    f32x16 acc[NT];
    {
      bf16x8 a_hi[KS], a_lo[KS];
      if constexpr (STAGE) {
        // 16-byte pieces per row (4 fp32 / 8 bf16 channels each) and per lane
        constexpr int EP = ABF ? 8 : 4, Q = C / EP, NL = 32 * Q / 64;
        f32x4 raw[NL][ABF ? 2 : 1];
#pragma unroll
        for (int k = 0; k < NL; ++k) {
          const int e = lane + 64 * k, pr = e / Q, q = e % Q;
          const int mm = wave * 32 + pr;
          const int gh = th0 + (mm >> lgTW), gw = tw0 + (mm & (TW - 1));
          const size_t o = (((size_t)nb * P.H + min(gh, P.H - 1)) * P.W + gw) * C + EP * q;
          const float okr = gh < P.H ? 1.0f : 0.0f;
          if (ABF) {
            float v8[8];
            act_ld8<ABF>(P.y, o, v8);
            raw[k][0] = f32x4{v8[0], v8[1], v8[2], v8[3]} * okr;
            raw[k][ABF ? 1 : 0] = f32x4{v8[4], v8[5], v8[6], v8[7]} * okr;
          } else {
            raw[k][0] = *reinterpret_cast<const f32x4*>(P.y + o) * okr;
          }
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");  // the previous tile's reads of the image are done
#pragma unroll
        for (int k = 0; k < NL; ++k) {
          const int e = lane + 64 * k;
          *reinterpret_cast<f32x4*>(Ys + (e / Q) * YROW + EP * (e % Q)) = raw[k][0];
          if (ABF) *reinterpret_cast<f32x4*>(Ys + (e / Q) * YROW + EP * (e % Q) + 4) = raw[k][ABF ? 1 : 0];
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");  // image complete before other lanes' rows are read
        const int mA = wave * 32 + li;
        const float okf = th0 + (mA >> lgTW) < P.H ? 1.0f : 0.0f;
#pragma unroll
        for (int ks = 0; ks < KS; ++ks) {
          const f32x4 r0 = *reinterpret_cast<const f32x4*>(Ys + li * YROW + 16 * ks + 8 * lh);
          const f32x4 r1 = *reinterpret_cast<const f32x4*>(Ys + li * YROW + 16 * ks + 8 * lh + 4);
          const float4 s0 = *reinterpret_cast<const float4*>(s_sc + 16 * ks + 8 * lh);
          const float4 s1 = *reinterpret_cast<const float4*>(s_sc + 16 * ks + 8 * lh + 4);
          const float4 h0 = *reinterpret_cast<const float4*>(s_sh + 16 * ks + 8 * lh);
          const float4 h1 = *reinterpret_cast<const float4*>(s_sh + 16 * ks + 8 * lh + 4);
          float v[8];
          v[0] = fmaf(r0[0], s0.x, h0.x) * okf; v[1] = fmaf(r0[1], s0.y, h0.y) * okf;
          v[2] = fmaf(r0[2], s0.z, h0.z) * okf; v[3] = fmaf(r0[3], s0.w, h0.w) * okf;
          v[4] = fmaf(r1[0], s1.x, h1.x) * okf; v[5] = fmaf(r1[1], s1.y, h1.y) * okf;
          v[6] = fmaf(r1[2], s1.z, h1.z) * okf; v[7] = fmaf(r1[3], s1.w, h1.w) * okf;
          split_pack8(v, a_hi[ks], a_lo[ks]);
        }
      } else {
        load_a_frags<C, ABF>(P, lgTW, s_sc, s_sh, nb, th0, tw0, wave, li, lh, a_hi, a_lo);
      }
#pragma unroll
      for (int j = 0; j < NT; ++j)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[j][r] = 0.f;
#pragma unroll
      for (int ks = 0; ks < KS; ++ks)
#pragma unroll
        for (int j = 0; j < NT; ++j) {
          const bf16x8 b_hi = WF[((j * KS + ks) * 2 + 0) * 64 + lane];
          const bf16x8 b_lo = WF[((j * KS + ks) * 2 + 1) * 64 + lane];
          acc[j] = mfma_sp<ABF>(a_hi[ks], a_lo[ks], b_hi, b_lo, acc[j]);
        }
    }

    // ---- gate + dropout in place: acc <- (lin + b) * sigmoid(xn) * mask/(1-p)   (rows below the image -> 0)
#pragma unroll
    for (int rg = 0; rg < 4; ++rg) {
      uint32_t yvr[4][NT];
      float mk[4];
      uint32_t posv[4];
#pragma unroll
      for (int rr = 0; rr < 4; ++rr) {
        const int mm = wave * 32 + 8 * rg + lhv + rr;
        const int gh = th0 + (mm >> lgTW), gw = tw0 + (mm & (TW - 1));
        mk[rr] = gh < P.H ? 1.0f : 0.0f;
        posv[rr] = ((uint32_t)(nb * P.H + min(gh, P.H - 1)) * (uint32_t)P.W + gw) * C + li;
#pragma unroll
        for (int j = 0; j < NT; ++j)
          yvr[rr][j] = STAGE ? __float_as_uint(Ys[(8 * rg + lhv + rr) * YROW + 32 * j + li]) : act_ld_raw<ABF>(P.y, posv[rr] + 32 * j);
      }
      if (!STAGE) __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int rr = 0; rr < 4; ++rr)
#pragma unroll
        for (int j = 0; j < NT; ++j) {
          const float xn = fmaf((STAGE ? __uint_as_float(yvr[rr][j]) : act_cvt<ABF>(yvr[rr][j])), csc[j], csh[j]);
          const float keep = mk[rr] * drop_mul32(posv[rr] + 32 * j, dkey, dthr, dscale);
          acc[j][4 * rg + rr] = (acc[j][4 * rg + rr] + bias[j]) * sigmoid_fast(xn) * keep;
        }
    }

    // ---- average pooling, lane-local
    if (P.ph == 1) {
      // windows (m, m+1) [pw = 2] or single positions [pw = 1]
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        if (P.pw == 2 && (r & 1)) continue;
        const int mm = wave * 32 + crow3g(r, 0) + lhv;
        const int gh = th0 + (mm >> lgTW), gw = tw0 + (mm & (TW - 1));
        const int gpw = gw >> spw;
        if (gh < P.H && gpw < P.Wp) {
          const size_t dst = (((size_t)nb * P.Hp + gh) * P.Wp + gpw) * C + li;
#pragma unroll
          for (int j = 0; j < NT; ++j) {
            const float v = P.pw == 2 ? acc[j][r] + acc[j][r | 1] : acc[j][r];
            act_st<ABF>(P.pooled, dst + 32 * j, v * inv_pool);
          }
        }
      }
    } else {
      // ph = 2: the partner row m + TW sits dr registers further (TW = 16 -> 8, TW = 8 -> 4, TW = 2 -> 2, TW = 1 -> 1)
      const int dr = TW == 16 ? 8 : (TW == 8 ? 4 : (TW == 2 ? 2 : 1));
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int mm = wave * 32 + crow3g(r, 0) + lhv;
        const int lr = mm >> lgTW;  // tile row
        const int gh = th0 + lr, gw = tw0 + (mm & (TW - 1));
        const int gph = gh >> 1, gpw = gw >> spw;
        const bool top = (lr & 1) == 0 && (P.pw == 1 || (gw & 1) == 0);
        if (top && gph < P.Hp && gpw < P.Wp) {
          const size_t dst = (((size_t)nb * P.Hp + gph) * P.Wp + gpw) * C + li;
#pragma unroll
          for (int j = 0; j < NT; ++j) {
            // (dynamic register offsets are resolved at compile time: r and dr take few values, selected below)
            float v;
            if (dr == 8) v = acc[j][r] + acc[j][(r + 8) & 15] + (P.pw == 2 ? acc[j][r | 1] + acc[j][((r + 8) & 15) | 1] : 0.f);
            else if (dr == 4) v = acc[j][r] + acc[j][(r + 4) & 15] + (P.pw == 2 ? acc[j][r | 1] + acc[j][((r + 4) & 15) | 1] : 0.f);
            else if (dr == 2) v = acc[j][r] + acc[j][(r + 2) & 15] + (P.pw == 2 ? acc[j][r | 1] + acc[j][((r + 2) & 15) | 1] : 0.f);
            else v = acc[j][r] + acc[j][(r + 1) & 15];  // TW = 1 (width-1 maps of the FPN levels): pw = 1
            act_st<ABF>(P.pooled, dst + 32 * j, v * inv_pool);
          }
        }
      }
    }
  }
}

template <int C, int ABF>
static int launch_glu_fwd3(const Glu3Params& P, int G, hipStream_t s) {
  constexpr int NT = C / 32, KS = C / 16;
  const size_t smem = (size_t)NT * KS * 2 * 64 * 16 + 2 * C * sizeof(float) +
                      (C <= 64 ? (size_t)4 * 32 * (C + 4) * sizeof(float) : 0);  // + the waves' staged y rows
  static BsedLdsOnce once, once4;
  if (glu3_const_tw(P)) {
    BSED_HIP(bsed_max_lds(once4, (const void*)glu_fwd3_kernel<C, 4, ABF>));
    hipLaunchKernelGGL((glu_fwd3_kernel<C, 4, ABF>), dim3(G), dim3(G3_THREADS), smem, s, P);
  } else {
    BSED_HIP(bsed_max_lds(once, (const void*)glu_fwd3_kernel<C, -1, ABF>));
    hipLaunchKernelGGL((glu_fwd3_kernel<C, -1, ABF>), dim3(G), dim3(G3_THREADS), smem, s, P);
  }
  BSED_LAUNCH_CHECK();
  return BSED_OK;
}

template <int C>
static size_t glu_bwd3_smem() {
  constexpr int NT = C / 32, KS = C / 16;
  return (size_t)2 * NT * KS * 2 * 64 * 16 + 2 * C * sizeof(float) + (size_t)4 * 32 * (2 * C + 8) * 2;
}

template <int C, int ABF>
static int launch_glu_bwd3(const Glu3Params& P, int G, hipStream_t s) {
  const size_t smem = glu_bwd3_smem<C>();
  static BsedLdsOnce once, once4;
  if (glu3_const_tw(P)) {
    BSED_HIP(bsed_max_lds(once4, (const void*)glu_bwd3_kernel<C, 4, ABF>));
    hipLaunchKernelGGL((glu_bwd3_kernel<C, 4, ABF>), dim3(G), dim3(G3_THREADS), smem, s, P);
  } else {
    BSED_HIP(bsed_max_lds(once, (const void*)glu_bwd3_kernel<C, -1, ABF>));
    hipLaunchKernelGGL((glu_bwd3_kernel<C, -1, ABF>), dim3(G), dim3(G3_THREADS), smem, s, P);
  }
  BSED_LAUNCH_CHECK();
  return BSED_OK;
}

static int fill_params(Glu3Params& P, int NB, int H, int W, int C, int TH, int TW, int ph, int pw, const char* who) {
  BSED_CHECK_ARG(NB > 0 && H > 0 && W > 0 && TH * TW == G3_M && W % TW == 0, "%s: bad shape", who);
  BSED_CHECK_ARG((ph == 1 || ph == 2) && (pw == 1 || pw == 2), "%s: pooling windows must be 1 or 2", who);
  P.NB = NB; P.H = H; P.W = W; P.TH = TH; P.TW = TW;
  P.lgTW = 0;
  while ((1 << P.lgTW) < TW) ++P.lgTW;
  BSED_CHECK_ARG((1 << P.lgTW) == TW, "%s: TW must be a power of two", who);
  P.tilesH = ceil_div(H, TH); P.tilesW = W / TW;
  const long ntiles = (long)NB * P.tilesH * P.tilesW;
  BSED_CHECK_ARG(ntiles < (1L << 31) && (long)NB * H * W * C < (1L << 32),
                 "%s: activation tensor beyond the 32-bit element offsets of this kernel", who);
  P.ntiles = (int)ntiles;
  P.ph = ph; P.pw = pw; P.Hp = H / ph; P.Wp = W / pw;
  return BSED_OK;
}

extern "C" int bsed_glu_bwd3_slabs(int C) { (void)C; return 4; }

// workgroups of one resident round (persistent kernel): LDS- and register-limited occupancy x 256 CUs
static int env_g(const char* name, int dflt) {   // A/B knob: workgroups of the persistent grid
  const char* v = getenv(name);
  return v && atoi(v) > 0 ? atoi(v) : dflt;
}
extern "C" int bsed_glu_bwd3_auto_g(int C) { return C == 32 ? env_g("BSED_GLU_BWD3_G32", 768) : env_g("BSED_GLU_BWD3_G64", 512); }

extern "C" int bsed_glu_bwd3(const float* y, const float* scale, const float* shift, const float* w, const float* bias,
                             const float* dpool, float* g, float* part_dw, float* part_db, float* part_st, int G,
                             int NB, int H, int W, int C, int TH, int TW, int ph, int pw, float drop_p,
                             uint32_t rng_stream, uint64_t seed, int act_bf16, void* stream) {
  BSED_CHECK_ARG(y && scale && shift && w && bias && dpool && g && part_dw && part_db && part_st,
                 "bsed_glu_bwd3: null tensor");
  BSED_CHECK_ARG(C == 32 || C == 64, "bsed_glu_bwd3: built for C in {32,64} (got %d)", C);
  Glu3Params P;
  int rc = fill_params(P, NB, H, W, C, TH, TW, ph, pw, "bsed_glu_bwd3");
  if (rc) return rc;
  BSED_CHECK_ARG(G > 0 && G <= P.ntiles, "bsed_glu_bwd3: G must be in 1..%d tiles", P.ntiles);
  P.y = y; P.scale = scale; P.shift = shift; P.w = w; P.bias = bias; P.dpool = dpool;
  P.g = g; P.part_dw = part_dw; P.part_db = part_db; P.part_st = part_st; P.pooled = nullptr;
  P.drop_p = drop_p; P.rng_stream = rng_stream; P.seed = seed; P.seed_add = bsed_seed_add_ptr();
  hipStream_t s = (hipStream_t)stream;
  if (act_bf16) return C == 64 ? launch_glu_bwd3<64, 1>(P, G, s) : launch_glu_bwd3<32, 1>(P, G, s);
  if (C == 64) return launch_glu_bwd3<64, 0>(P, G, s);
  return launch_glu_bwd3<32, 0>(P, G, s);
}

extern "C" int bsed_glu_fwd3_auto_g(int C) {
  // measured (B = 256): C = 32: 512 / 768 / 1024 / 1536 workgroups 0.447 / 0.367 / 0.412 / 0.350 ms; C = 64: 0.180 / 0.160 / 0.167 / 0.155;
  // C = 128 (64 KB of weight fragments in LDS, two per CU): 256 / 512 / 768 / 1024 0.514 / 0.422 / 0.454 / 0.435
  return C == 32 ? env_g("BSED_GLU_FWD3_G32", 1536) : C == 64 ? env_g("BSED_GLU_FWD3_G64", 1536) : env_g("BSED_GLU_FWD3_G128", 512);
}

extern "C" int bsed_glu_fwd3(const float* y, const float* scale, const float* shift, const float* w, const float* bias,
                             float* pooled, int G, int NB, int H, int W, int C, int TH, int TW, int ph, int pw,
                             float drop_p, uint32_t rng_stream, uint64_t seed, int act_bf16, void* stream) {
  BSED_CHECK_ARG(y && scale && shift && w && bias && pooled, "bsed_glu_fwd3: null tensor");
  BSED_CHECK_ARG(C == 32 || C == 64 || C == 128, "bsed_glu_fwd3: built for C in {32,64,128} (got %d)", C);
  Glu3Params P;
  int rc = fill_params(P, NB, H, W, C, TH, TW, ph, pw, "bsed_glu_fwd3");
  if (rc) return rc;
  BSED_CHECK_ARG(G > 0 && G <= P.ntiles, "bsed_glu_fwd3: G must be in 1..%d tiles", P.ntiles);
  BSED_CHECK_ARG(pw == 1 || TW >= 2, "bsed_glu_fwd3: horizontal pooling needs TW >= 2");
  BSED_CHECK_ARG(ph == 1 || ((TW == 16 || TW == 8 || TW == 2 || (TW == 1 && pw == 1)) && TH % 2 == 0),
                 "bsed_glu_fwd3: vertical pooling is lane-local only for TW in {1,2,8,16} (got %d)", TW);
  P.y = y; P.scale = scale; P.shift = shift; P.w = w; P.bias = bias; P.dpool = nullptr;
  P.g = nullptr; P.part_dw = nullptr; P.part_db = nullptr; P.part_st = nullptr; P.pooled = pooled;
  P.drop_p = drop_p; P.rng_stream = rng_stream; P.seed = seed; P.seed_add = bsed_seed_add_ptr();
  hipStream_t s = (hipStream_t)stream;
  if (act_bf16) return C == 128 ? launch_glu_fwd3<128, 1>(P, G, s) : (C == 64 ? launch_glu_fwd3<64, 1>(P, G, s) : launch_glu_fwd3<32, 1>(P, G, s));
  if (C == 128) return launch_glu_fwd3<128, 0>(P, G, s);
  if (C == 64) return launch_glu_fwd3<64, 0>(P, G, s);
  return launch_glu_fwd3<32, 0>(P, G, s);
}

// C = 128: g, d_lin (for the separate weight-gradient pass), db and BatchNorm-backward partials.  frag_table: 128 KB
// of device scratch that this call fills (bsed_glu_bwd3n_table_bytes()).
extern "C" size_t bsed_glu_bwd3n_table_bytes(void) { return (size_t)2 * 4 * 8 * 2 * 64 * 16; }

extern "C" int bsed_glu_bwd3n(const float* y, const float* scale, const float* shift, const float* w, const float* bias,
                              const float* dpool, float* g, float* dlin, float* part_db, float* part_st,
                              void* frag_table, int G, int NB, int H, int W, int C, int TH, int TW, int ph, int pw,
                              float drop_p, uint32_t rng_stream, uint64_t seed, int act_bf16, void* stream) {
  BSED_CHECK_ARG(y && scale && shift && w && bias && dpool && g && dlin && part_db && part_st && frag_table,
                 "bsed_glu_bwd3n: null tensor");
  BSED_CHECK_ARG(C == 128, "bsed_glu_bwd3n: built for C = 128 (got %d)", C);
  Glu3Params P;
  int rc = fill_params(P, NB, H, W, C, TH, TW, ph, pw, "bsed_glu_bwd3n");
  if (rc) return rc;
  BSED_CHECK_ARG(G > 0 && 2 * (G - 1) < P.ntiles, "bsed_glu_bwd3n: G workgroups take two tiles at a time: 1..%d",
                 (P.ntiles + 1) / 2);
  P.y = y; P.scale = scale; P.shift = shift; P.w = w; P.bias = bias; P.dpool = dpool;
  P.g = g; P.part_dw = nullptr; P.part_db = part_db; P.part_st = part_st; P.pooled = nullptr;
  P.drop_p = drop_p; P.rng_stream = rng_stream; P.seed = seed; P.seed_add = bsed_seed_add_ptr();
  hipStream_t s = (hipStream_t)stream;
  hipLaunchKernelGGL(glu3_pack_frags_kernel, dim3(4 * 8 * 64 / 256), dim3(256), 0, s, w, (bf16x8*)frag_table);
  const size_t smem = bsed_glu_bwd3n_table_bytes() + 2 * 128 * sizeof(float) + (size_t)8 * 32 * (2 * 16 + 8) * 2;
  static BsedLdsOnce once, onceb;
  if (act_bf16) {
    BSED_HIP(bsed_max_lds(onceb, (const void*)glu_bwd3n_kernel<1>));
    hipLaunchKernelGGL(glu_bwd3n_kernel<1>, dim3(G), dim3(G3N_THREADS), smem, s, P, (const bf16x8*)frag_table, dlin);
  } else {
    BSED_HIP(bsed_max_lds(once, (const void*)glu_bwd3n_kernel<0>));
    hipLaunchKernelGGL(glu_bwd3n_kernel<0>, dim3(G), dim3(G3N_THREADS), smem, s, P, (const bf16x8*)frag_table, dlin);
  }
  BSED_LAUNCH_CHECK();
  return BSED_OK;
}
