// 3x3 convolution forward / data-gradient, split-fp32 ("bf16x3") implicit GEMM, round-3 structure: N-SPLIT waves,
// weight fragments straight from global memory, no per-tap barriers.
//
// What igemm3_kernel (igemm3.hip) pays per (32-channel chunk, tap) step: a weight slab written to LDS by all four
// waves, TWO workgroup barriers, and 20 ds_read_b128 per 24 MFMAs; its co-resident workgroups then run the same
// stage -> barrier -> read -> multiply program in phase (skeleton 51 % + MFMA 49 % of the kernel
// time ADD).  Here a wave owns 32 output channels and ALL the positions of its row blocks:
//   * its B operand (the weights of those 32 channels) is private to the wave, so it never goes through LDS: the
//     weights are packed in MFMA fragment order (bsed_pack_weight3s layout, any K) and a step's four fragments
//     (k halves x hi / lo) are four coalesced 1 KB global loads per wave, issued one step ahead into registers
//     (L2 / L1 resident: the whole layer's table is 590 KB);
//   * only the activation patch lives in LDS, double-buffered per 32-channel chunk: ONE barrier per chunk (9 taps)
//     instead of 18, and between barriers the waves drift freely, so reads, MFMAs and epilogues of different waves
//     overlap by themselves;
//   * BatchNorm partial sums need no cross-wave reduction (a channel belongs to one wave per row group): no barrier
//     and no LDS in the epilogue; a tile writes 4 / NWN partial rows.
// Accumulation order per output element is the one of igemm3_kernel (chunk, tap, k half, lo*hi, hi*lo, hi*hi): the
// output tensor is bit-identical; the statistics rows are summed in a different (fixed) order.
#include "bsed_common.h"
#include "../../include/bsed.h"
#include <algorithm>
#include <atomic>
#include <stdlib.h>
#include <type_traits>

#define I3N_THREADS 256
#define I3N_M 128
#define I3N_KC 32
#define I3N_ROW 72  // ushorts per LDS patch row: 32 hi + 32 lo + 8 pad (144 B: conflict-free 16-byte fragment reads)

typedef short bf16x8 __attribute__((ext_vector_type(8)));
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));

struct Igemm3nParams {
  BsedIgemmDesc d;
  int PW, PH, PP, lgTW, b_off, pw_magic, prio;
};

// Diagnostic build (-DI3N_STAMP, tools/build_variant.sh): s_memtime stamps of one wave's phases go to the buffer passed
// in desc.e_src ([tile][wave][20] uint64, 16 / 17 = s_memrealtime at start / end, 18 / 19 = HW_ID / XCC_ID; no output value depends on them)
#ifdef I3N_STAMP
#define I3N_T(i)                                                                                         \
  do {                                                                                                   \
    __builtin_amdgcn_sched_barrier(0);                                                                   \
    unsigned long long t__;                                                                              \
    asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t__)::"memory");                          \
    __builtin_amdgcn_sched_barrier(0);                                                                   \
    if (lane == 0) stampbuf[i] = t__;                                                                    \
  } while (0)
#else
#define I3N_T(i)
#endif

__device__ __forceinline__ int crow3n(int r, int lh) { return (r & 3) + 8 * (r >> 2) + 4 * lh; }

// NWN = waves along N (each 32 channels): 4 -> BN = 128, a wave covers all four 32-row blocks of the tile;
// 2 -> BN = 64, two row blocks; 1 -> BN = 32, one row block (the four waves share the B fragments through L1).
// NT9 = 1: nine taps, loop fully unrolled (tap offsets and the alternating B buffers become static).
// MW = waves along M: NWN * MW = 4 waves (256 threads), or -- NWN = 4, MW = 2 -- 8 waves (512 threads: a wave covers
// two row blocks, 32 accumulator registers, four waves per SIMD; measured slower -- twice the weight-fragment loads --
// and not instantiated).  P.prio: raised wave priority outside the MFMA loop (prologue and epilogue are issue-bound
// beside two waves that multiply: 1-3 % on the PV = 9 layers).
// ABF = 1: the "bf16" throughput mode (BASELINE configs[1-2]): input AND output activations are bf16 in HBM, ONE bf16
// MFMA per product (the hi halves of the same weight table), fp32 accumulation, bias and BatchNorm sums; the patch
// goes to LDS as it arrives (rows of 32 bf16 + 8 pad = 80 B: 20 r mod 64 hits 16 distinct 4-bank groups).
// SH = 1: the same contraction on v_mfma_f32_16x16x32_bf16 tiles (a 32-row block = two row tiles, a wave's 32 channels =
// two column tiles, K = 32 = one chunk per MFMA): same MFMA cycles and LDS reads per step, twice the MFMA instructions.
// 8-16 % faster per layer at two waves per SIMD; in-kernel stamps show the SAME clock for both shapes (1.82 GHz) and a
// shorter wave life in cycles (82.7 k -> 72.9 k on the 216 x 8 layer): twice as many, half as long, independent MFMAs per
// step keep the matrix pipe fed better around the LDS reads.
// The sums run over K in a different order (32 per MFMA instead of 2 x 16): results agree with SH = 0 to rounding.
template <int NWN, int MW, int STATS, int PV, int NT9, int WPE, int ABF, int SH = 0>
__global__ __launch_bounds__(64 * NWN * MW) __attribute__((amdgpu_waves_per_eu(WPE, WPE)))
void igemm3n_kernel(const Igemm3nParams P) {
  constexpr int RB = 4 / MW, NTH = 64 * NWN * MW;
  constexpr int ROW = ABF ? 40 : I3N_ROW;      // ushorts per LDS patch row
  constexpr int LGP = ABF ? 2 : 3;             // log2 of 16-byte pieces per patch position (8 bf16 / 4 fp32 channels each)
  constexpr int NF = ABF ? 2 : 4;              // weight fragments per step
  const bool PRIO = P.prio != 0;
  if (PRIO) __builtin_amdgcn_s_setprio(3);
  const BsedIgemmDesc& p = P.d;
  extern __shared__ __align__(16) unsigned short smem3n[];
  const int tid = threadIdx.x, lane = tid & 63, li = lane & 31, lh = lane >> 5;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);   // wave-uniform by construction: keep it in an SGPR
  const int wn = wave % NWN, wm = wave / NWN;
  // XCD-affine tile order: workgroups go round-robin over the 8 XCDs, neighbouring tiles (shared halo rows) should
  // meet in one L2 (bijective for any grid size; speed only)
  int tile;
  {
    const int nt = gridDim.x, bid = blockIdx.x, q = nt >> 3, r = nt & 7, x = bid & 7;
    tile = (x < r ? x * (q + 1) : r * (q + 1) + (x - r) * q) + (bid >> 3);
  }
#ifdef I3N_STAMP
  unsigned long long* stampbuf = (unsigned long long*)p.e_src + ((size_t)tile * 4 + wave) * 20;
#endif
  I3N_T(0);
#ifdef I3N_STAMP
  if (lane == 0) stampbuf[16] = __builtin_amdgcn_s_memrealtime();
  {
    unsigned hwid, xcc;
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hwid));
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
    if (lane == 0) { stampbuf[18] = hwid; stampbuf[19] = xcc; }
  }
#endif
  const int tw_i = tile % p.tilesW; tile /= p.tilesW;
  const int th_i = tile % p.tilesH;
  const int nb = tile / p.tilesH;
  const int th0 = th_i * p.TH, tw0 = tw_i * p.TW;
  const int n0 = blockIdx.y * (32 * NWN);
  const int PW = P.PW;
  int abase[RB];
#pragma unroll
  for (int rb = 0; rb < RB; ++rb) {
    const int m = (wm * RB + rb) * 32 + li;
    abase[rb] = (((m >> P.lgTW) + p.hh) * PW + (m & (p.TW - 1)) + p.hw) * ROW + 8 * lh;
  }
  f32x16 acc[RB];
#pragma unroll
  for (int rb = 0; rb < RB; ++rb)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[rb][r] = 0.f;
  // SH = 1: lane (li16 = row of the 16-row tile / column of the 16-column tile, kg = 8-channel group of the chunk)
  const int li16 = lane & 15, kg = lane >> 4;
  int abase16[SH ? RB : 1][2];
  f32x4 acc16[SH ? RB : 1][2][2];   // [row block][row tile][column tile]
  if (SH) {
#pragma unroll
    for (int rb = 0; rb < RB; ++rb)
#pragma unroll
      for (int h = 0; h < 2; ++h) {
        const int m = (wm * RB + rb) * 32 + 16 * h + li16;
        abase16[rb][h] = (((m >> P.lgTW) + p.hh) * PW + (m & (p.TW - 1)) + p.hw) * ROW + 8 * kg;
#pragma unroll
        for (int ct = 0; ct < 2; ++ct) acc16[rb][h][ct] = f32x4{0.f, 0.f, 0.f, 0.f};
      }
  }

  // 16-byte units of the input image: 4 fp32 / 8 bf16 channels; poff counts them
  const u32x4* inb = reinterpret_cast<const u32x4*>(reinterpret_cast<const char*>(p.in) +
                                                    (size_t)nb * p.H * p.W * p.in_pitch * (ABF ? 2 : 4));
  const int nchunks = p.CIN / I3N_KC, KS = 2 * nchunks, ntaps = NT9 ? 9 : p.ntaps;
  const int a_total = P.PP << LGP;
  const int pitch16 = p.in_pitch >> (ABF ? 3 : 2), chunk16 = 1 << LGP;   // 16-byte units per position / per chunk
  u32x4 pv[PV];
  int poff[PV];
#pragma unroll
  for (int u = 0; u < PV; ++u) {
    const int e = tid + u * NTH;
    const int c4 = e & (chunk16 - 1), pos = e >> LGP;
    const int pr = (pos * P.pw_magic) >> 20, pc = pos - pr * PW;
    const int gh = th0 - p.hh + pr, gw = tw0 - p.hw + pc;
    poff[u] = (e < a_total && gh >= 0 && gh < p.H && gw >= 0 && gw < p.W) ? (gh * p.W + gw) * pitch16 + c4 : -1;
  }
  // branch-free loads (a divergent branch around a load turns every later counted wait into vmcnt(0)): padding
  // positions read the image's first element and are zeroed when the patch is written to LDS
#pragma unroll
  for (int u = 0; u < PV; ++u) pv[u] = inb[max(poff[u], 0)];

  // bias of this lane's channel: requested here so that the epilogue never waits for it
  const int n = n0 + 32 * wn + li;
  const bool nok = n < p.N;
  const float bias = (p.bias && nok) ? p.bias[n] : 0.f;
  // SH = 1: the lane's two channels (one per column tile)
  const int n16 = n0 + 32 * wn + li16;
  float bias16[2] = {0.f, 0.f};
  if (SH && p.bias) {
    if (n16 < p.N) bias16[0] = p.bias[n16];
    if (n16 + 16 < p.N) bias16[1] = p.bias[n16 + 16];
  }

  // this wave's weight fragments: table[jn][tap][k16][hi|lo][lane] of 16-byte elements (bsed_pack_weight3s layout)
  // SH = 1: column tile ct of the wave's 32 channels; the lane's eight k values (8 kg ..) of column 16 ct + li16 sit in
  // the table's 32 x 32 x 16 fragment of k step kg >> 1 at lane (li = 16 ct + li16, lh = kg & 1): fragment index
  // f = [ct][hi | lo] (ABF: [ct]) at offset ((kg >> 1) * 2 + hl) * 64 + 16 ct from a per-lane base
  const u32x4* wb = reinterpret_cast<const u32x4*>(p.w) + ((size_t)(n0 / 32 + wn) * ntaps * KS) * 128 +
                    (SH ? (kg >> 1) * 128 + (kg & 1) * 32 + li16 : lane);
  auto foff = [&](int f) { return SH ? (ABF ? 0 : (f & 1)) * 64 + 16 * (ABF ? f : f >> 1) : (ABF ? 2 * f : f) * 64; };
  u32x4 bq[NF];   // the current step's fragments: [k half][hi | lo]  (ABF: [k half], hi only); SH: [column tile][hi | lo]
#pragma unroll
  for (int f = 0; f < NF; ++f) bq[f] = wb[foff(f)];

  // piece u of the prefetched patch chunk -> LDS (split into bf16 hi / lo; padding positions become zeros)
  auto write_piece = [&](int u, unsigned short* dstbuf) {
    const int e = tid + u * NTH;
    if (e < a_total) {
      if (ABF) {
        const u32x4 v = poff[u] >= 0 ? pv[u] : u32x4{0u, 0u, 0u, 0u};
        *reinterpret_cast<u32x4*>(dstbuf + (e >> 2) * ROW + 8 * (e & 3)) = v;
      } else {
        uint32_t h01, l01, h23, l23;
        const f32x4 v = poff[u] >= 0 ? __builtin_bit_cast(f32x4, pv[u]) : f32x4{0.f, 0.f, 0.f, 0.f};
        bsed_split2(v[0], v[1], h01, l01);
        bsed_split2(v[2], v[3], h23, l23);
        unsigned short* dst = dstbuf + (e >> 3) * ROW + 4 * (e & 7);
        *reinterpret_cast<uint2*>(dst) = make_uint2(h01, h23);
        *reinterpret_cast<uint2*>(dst + 32) = make_uint2(l01, l23);
      }
    }
  };
  // NT9: the patch of chunk ch + 1 is written into the OTHER buffer during the taps of chunk ch (one piece per tap
  // from tap 3 on: its loads, issued at the top of the chunk, have landed by then), so that a wave reaches the
  // chunk's barrier with nothing left to do; otherwise (1 - 4 taps) the whole chunk is written at the top of its
  // own iteration.  Either way: ONE barrier per chunk.  Buffer (ch + 1) & 1 was last read during chunk ch - 1, and
  // every wave has passed the barrier at the top of chunk ch since.
  constexpr bool EARLYW = NT9 != 0;
  I3N_T(1);
  if (EARLYW) {
#pragma unroll
    for (int u = 0; u < PV; ++u) write_piece(u, smem3n);
  }
  auto chunk = [&](int ch, auto more_c) {
    constexpr bool more = decltype(more_c)::value;   // a next chunk exists (the last chunk is peeled off the loop)
    unsigned short* As = smem3n + (ch & 1) * P.b_off;
    unsigned short* An = smem3n + ((ch + 1) & 1) * P.b_off;
    if (!EARLYW) {
#pragma unroll
      for (int u = 0; u < PV; ++u) write_piece(u, As);
    }
    if (ch < 4) I3N_T(2 + 3 * ch);
    __syncthreads();   // the ONE barrier of the chunk: patch visible; everyone is out of chunk ch - 1
    if (ch < 4) I3N_T(3 + 3 * ch);
    if (PRIO && ch == 0) __builtin_amdgcn_s_setprio(0);
    if (!EARLYW && more) {
#pragma unroll
      for (int u = 0; u < PV; ++u) pv[u] = inb[max(poff[u], 0) + (ch + 1) * chunk16];
    }
    auto step = [&](int tap) {
      // next step's fragments (the very last step re-reads its own: no branch around the loads)
      int ntap = tap + 1, nch = ch;
      if (ntap == ntaps) { ntap = 0; nch = ch + 1; }
      if (!more && nch != ch) { ntap = tap; nch = ch; }
      const u32x4* wn_ = wb + (size_t)(ntap * KS + 2 * nch) * 128;
      u32x4 bn[NF];
#pragma unroll
      for (int f = 0; f < NF; ++f) bn[f] = wn_[foff(f)];
      // (hipcc sinks these loads to just before their first use -- a step later -- to save registers, which exposes
      //  the L2 latency twice per step; the scheduling fence below keeps them ahead of this step's reads and MFMAs)
      if (EARLYW && more) {
        // piece u of the next chunk: requested at tap 6 u / PV (0..5), written three taps later (3..8): at most
        // half the pieces are in registers at a time
#pragma unroll
        for (int u = 0; u < PV; ++u)
          if (tap == (6 * u) / PV) pv[u] = inb[max(poff[u], 0) + (ch + 1) * chunk16];
      }
      __builtin_amdgcn_sched_barrier(0);
      if (EARLYW && more) {
#pragma unroll
        for (int u = 0; u < PV; ++u)
          if (tap == (6 * u) / PV + 3) write_piece(u, An);
      }
      const int toff = (p.dh[tap] * PW + p.dw[tap]) * ROW;
      if constexpr (SH) {
#pragma unroll
        for (int rb = 0; rb < RB; ++rb) {
          bf16x8 a_hi[2], a_lo[2];
#pragma unroll
          for (int h = 0; h < 2; ++h) {
            a_hi[h] = *reinterpret_cast<const bf16x8*>(As + abase16[rb][h] + toff);
            if (!ABF) a_lo[h] = *reinterpret_cast<const bf16x8*>(As + abase16[rb][h] + toff + 32);
          }
#pragma unroll
          for (int ct = 0; ct < 2; ++ct) {
            const bf16x8 b_hi = __builtin_bit_cast(bf16x8, bq[ABF ? ct : 2 * ct]);
            if (!ABF) {
              const bf16x8 b_lo = __builtin_bit_cast(bf16x8, bq[ABF ? ct : 2 * ct + 1]);
#pragma unroll
              for (int h = 0; h < 2; ++h) acc16[rb][h][ct] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a_lo[h], b_hi, acc16[rb][h][ct], 0, 0, 0);
#pragma unroll
              for (int h = 0; h < 2; ++h) acc16[rb][h][ct] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a_hi[h], b_lo, acc16[rb][h][ct], 0, 0, 0);
            }
#pragma unroll
            for (int h = 0; h < 2; ++h) acc16[rb][h][ct] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a_hi[h], b_hi, acc16[rb][h][ct], 0, 0, 0);
          }
        }
      } else
#pragma unroll
      for (int kk = 0; kk < 2; ++kk) {
        bf16x8 a_hi[RB], a_lo[RB];
#pragma unroll
        for (int rb = 0; rb < RB; ++rb) {
          a_hi[rb] = *reinterpret_cast<const bf16x8*>(As + abase[rb] + toff + 16 * kk);
          if (!ABF) a_lo[rb] = *reinterpret_cast<const bf16x8*>(As + abase[rb] + toff + 32 + 16 * kk);
        }
        const bf16x8 b_hi = __builtin_bit_cast(bf16x8, bq[ABF ? kk : 2 * kk]);
        if (!ABF) {
          const bf16x8 b_lo = __builtin_bit_cast(bf16x8, bq[ABF ? kk : 2 * kk + 1]);
#pragma unroll
          for (int rb = 0; rb < RB; ++rb) acc[rb] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a_lo[rb], b_hi, acc[rb], 0, 0, 0);
#pragma unroll
          for (int rb = 0; rb < RB; ++rb) acc[rb] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a_hi[rb], b_lo, acc[rb], 0, 0, 0);
        }
#pragma unroll
        for (int rb = 0; rb < RB; ++rb) acc[rb] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a_hi[rb], b_hi, acc[rb], 0, 0, 0);
      }
#pragma unroll
      for (int f = 0; f < NF; ++f) bq[f] = bn[f];
    };
    if (NT9) {
#pragma unroll
      for (int tap = 0; tap < 9; ++tap) step(tap);
    } else {
      for (int tap = 0; tap < ntaps; ++tap) step(tap);
    }
    if (ch < 4) I3N_T(4 + 3 * ch);
  };
  for (int ch = 0; ch < nchunks - 1; ++ch) chunk(ch, std::true_type{});
  chunk(nchunks - 1, std::false_type{});

  // ---- epilogue: + bias, store, optional BatchNorm partial sums (per wave: its 32 channels over its row blocks).
  // Address arithmetic is the cost here, not the stores (the slab kernel's epilogue computed a 64-bit address with
  // four integer multiplies per element: ~30 instructions x 64 stores = a quarter of a wave's life, in-kernel
  // stamps): one UNIFORM 64-bit tile origin, a 32-bit offset per group of four consecutive positions, uniform
  // increments inside the group, and no per-element bounds test on interior tiles.
  if (PRIO) __builtin_amdgcn_s_setprio(3);
  const int vh = p.valid_h > 0 ? p.valid_h : p.H, vw = p.valid_w > 0 ? p.valid_w : p.W;
  const bool full = th0 + p.TH <= vh && tw0 + p.TW <= vw && n0 + 32 * NWN <= p.N;
  // position m of the tile = m_u (uniform: row block, row group) + 4 lh (lane) + q, three disjoint bit ranges, so
  // its (row, column) in the tile -- and with it the element offset -- is the SUM of the three parts' offsets:
  // a scalar base per store (SALU), ONE per-lane byte offset for all 16 RB stores, no vector address arithmetic
  auto eoff = [&](int m) { return ((m >> P.lgTW) * p.W + (m & (p.TW - 1))) * p.out_pitch; };
  constexpr uint32_t OSZ = ABF ? 2u : 4u;   // bytes per output element
  char* ob = reinterpret_cast<char*>(p.out) + (((size_t)nb * p.H + th0) * p.W + tw0) * p.out_pitch * OSZ;
  const uint32_t voff = (uint32_t)(eoff(4 * lh) + n) * OSZ;
  auto put = [&](char* dst, float v) {
    if (ABF) *reinterpret_cast<__bf16*>(dst) = (__bf16)v;   // round to nearest even (v_cvt_pk_bf16_f32)
    else *reinterpret_cast<float*>(dst) = v;
  };
  float s0 = 0.f, s1 = 0.f;
  if constexpr (SH) {
    // result fragment of a 16 x 16 tile: column li16, rows 4 kg + r.  Position m = (32 (wm RB + rb) + 16 h) [uniform]
    // + 4 kg [lane] + r: three disjoint bit ranges again -- a scalar base per store, one per-lane byte offset
    const uint32_t voff16 = (uint32_t)(eoff(4 * kg) + n16) * OSZ;
    float t0[2] = {0.f, 0.f}, t1[2] = {0.f, 0.f};
    auto emit = [&](auto guarded_c) {
      constexpr bool guarded = decltype(guarded_c)::value;   // interior tiles store unmasked: no per-element test
#pragma unroll
      for (int rb = 0; rb < RB; ++rb)
#pragma unroll
        for (int h = 0; h < 2; ++h)
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            const int mu = (wm * RB + rb) * 32 + 16 * h;
            char* sb = ob + (size_t)(uint32_t)(eoff(mu) + eoff(r)) * OSZ;
            const int mm = mu + 4 * kg + r;
            const bool pok = !guarded || (th0 + (mm >> P.lgTW) < vh && tw0 + (mm & (p.TW - 1)) < vw);
#pragma unroll
            for (int ct = 0; ct < 2; ++ct) {
              const float v = acc16[rb][h][ct][r] + bias16[ct];
              if (!guarded || (pok && n16 + 16 * ct < p.N)) {
                put(sb + voff16 + 16 * ct * OSZ, v);
                if (STATS) { t0[ct] += v; t1[ct] = fmaf(v, v, t1[ct]); }
              }
            }
          }
    };
    if (full) emit(std::false_type{}); else emit(std::true_type{});
    if (STATS) {
      // the four row groups (kg) of a column: lanes li16, li16 + 16, + 32, + 48
#pragma unroll
      for (int ct = 0; ct < 2; ++ct) {
        float a = t0[ct] + __shfl_xor(t0[ct], 16, 64), b = t1[ct] + __shfl_xor(t1[ct], 16, 64);
        a += __shfl_xor(a, 32, 64); b += __shfl_xor(b, 32, 64);
        if (kg == 0 && n16 + 16 * ct < p.N) {
          const size_t row = ((size_t)(nb * p.tilesH + th_i) * p.tilesW + tw_i) * MW + wm;
          p.stats[(row * 2 + 0) * p.N + n16 + 16 * ct] = a;
          p.stats[(row * 2 + 1) * p.N + n16 + 16 * ct] = b;
        }
      }
    }
  } else if (full) {
#pragma unroll
    for (int rb = 0; rb < RB; ++rb)
#pragma unroll
      for (int rg = 0; rg < 4; ++rg)
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          const float v = acc[rb][4 * rg + q] + bias;
          char* sb = ob + (size_t)(uint32_t)(eoff((wm * RB + rb) * 32 + 8 * rg) + eoff(q)) * OSZ;
          put(sb + voff, v);
          if (STATS) { s0 += v; s1 = fmaf(v, v, s1); }
        }
  } else {
#pragma unroll
    for (int rb = 0; rb < RB; ++rb)
#pragma unroll
      for (int rg = 0; rg < 4; ++rg) {
        const int mm0 = (wm * RB + rb) * 32 + 8 * rg + 4 * lh;
        const int dr0 = mm0 >> P.lgTW, dc0 = mm0 & (p.TW - 1);
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          const float v = acc[rb][4 * rg + q] + bias;
          if (nok && th0 + dr0 + (q >> P.lgTW) < vh && tw0 + dc0 + (q & (p.TW - 1)) < vw) {
            char* sb = ob + (size_t)(uint32_t)(eoff((wm * RB + rb) * 32 + 8 * rg) + eoff(q)) * OSZ;
            put(sb + voff, v);
            if (STATS) { s0 += v; s1 = fmaf(v, v, s1); }
          }
        }
      }
  }
  if (STATS && !SH) {
    const float a = s0 + __shfl_xor(s0, 32, 64), b = s1 + __shfl_xor(s1, 32, 64);
    if (lh == 0 && nok) {
      // row (tile, wm) of the partial-sum table; the tile index here is the LOGICAL one (any fixed order will do)
      const size_t row = ((size_t)(nb * p.tilesH + th_i) * p.tilesW + tw_i) * MW + wm;
      p.stats[(row * 2 + 0) * p.N + n] = a;
      p.stats[(row * 2 + 1) * p.N + n] = b;
    }
  }
  I3N_T(14);
#ifdef I3N_STAMP
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  I3N_T(15);
  if (lane == 0) stampbuf[17] = __builtin_amdgcn_s_memrealtime();
#endif
}

// MFMA shape of the nine-tap instances (and of the two-wave BN = 128 builds of the 1 - 4 tap forms): 16 = 16 x 16 x 32 (default with fp32 activations: 8-16 % faster per layer,
// tools/conv_ab.py), 32 = 32 x 32 x 16 (default with bf16 activations, where the 16 form's extra registers spill at
// three waves per SIMD and the step is not matrix-bound).  BSED_IGEMM3N_SHAPE=16 / 32 forces one (A/B knob).
static std::atomic<int> i3n_forced_shape{-1};
extern "C" void bsed_igemm3n_set_shape(int shape) { i3n_forced_shape.store(shape == 16 || shape == 32 ? shape : 0); }
static int i3n_shape16(int abf) {
  int v = i3n_forced_shape.load();
  if (v < 0) {
    v = getenv("BSED_IGEMM3N_SHAPE") ? atoi(getenv("BSED_IGEMM3N_SHAPE")) : 0;
    i3n_forced_shape.store(v);
  }
  return v ? v == 16 : !abf;
}
template <int NWN, int MW, int STATS, int PV, int NT9, int WPE, int ABF = 0>
static int launch_i3n6(const Igemm3nParams& P, dim3 grid, size_t smem, hipStream_t s) {
  if constexpr (NT9 == 1 || (NWN == 4 && WPE == 2)) {   // (... and the two-wave BN = 128 builds of the 1 - 4 tap forms: GRU projections)
    if (i3n_shape16(ABF)) {
      static BsedLdsOnce once16;
      BSED_HIP(bsed_max_lds(once16, (const void*)igemm3n_kernel<NWN, MW, STATS, PV, NT9, WPE, ABF, 1>));
      hipLaunchKernelGGL((igemm3n_kernel<NWN, MW, STATS, PV, NT9, WPE, ABF, 1>), grid, dim3(64 * NWN * MW), smem, s, P);
      BSED_LAUNCH_CHECK();
      return BSED_OK;
    }
  }
  static BsedLdsOnce once;
  BSED_HIP(bsed_max_lds(once, (const void*)igemm3n_kernel<NWN, MW, STATS, PV, NT9, WPE, ABF>));
  hipLaunchKernelGGL((igemm3n_kernel<NWN, MW, STATS, PV, NT9, WPE, ABF>), grid, dim3(64 * NWN * MW), smem, s, P);
  BSED_LAUNCH_CHECK();
  return BSED_OK;
}

// A/B knob (bsed_igemm3n_set_wpe, BSED_IGEMM3N_WPE): 2 / 3 = the BN = 128 build for that many waves per SIMD (256 / 168
// registers) whatever the shape; + 8 = no raised priority outside the loop; 0 = default
static std::atomic<int> i3n_forced_wpe{-1};
extern "C" void bsed_igemm3n_set_wpe(int wpe) { i3n_forced_wpe.store(wpe); }
static int i3n_knob() {
  int forced = i3n_forced_wpe.load();
  if (forced < 0) {
    forced = getenv("BSED_IGEMM3N_WPE") ? atoi(getenv("BSED_IGEMM3N_WPE")) : 0;
    i3n_forced_wpe.store(forced);
  }
  return forced;
}

// which build runs a (BN, patch size, taps) combination (PV = 0: unsupported patch size)
struct I3nPlan { int NWN, MW, PV, WPE; };
static I3nPlan i3n_plan(int NP, int PP, int ntaps, int abf) {
  const int NWN = NP % 128 == 0 ? 4 : (NP % 64 == 0 ? 2 : 1);
  const int wpe = i3n_knob() & 7;
  I3nPlan pl{NWN, 4 / NWN, 0, 3};
  if (abf) {   // bf16 activations: 16-byte pieces of 8 channels, half the registers: three waves per SIMD throughout
    const int need = ceil_div(PP * 4, 256);
    pl.PV = need <= 3 ? 3 : (need <= 5 ? 5 : 0);
    return pl;
  }
  const int need = ceil_div(PP * 8, 256);
  pl.PV = need <= 6 ? 6 : (need <= 9 ? 9 : 0);
  // BN = 128: three waves per SIMD (168 registers) only for the shape that fits them without spilling in the loop
  // (nine taps, patch of <= 192 positions): 339 vs 359 us on the 216 x 8 layer; two otherwise (PV = 9: 173 vs 192 us
  // on the 216 x 4 layer).  tools/conv_ab.py
  // (the 16 x 16 x 32 form needs ~185 registers at BN = 128: two waves per SIMD, where it is 8-12 % faster than the
  //  32 x 32 x 16 form at three)
  if (NWN == 4) pl.WPE = wpe == 2 || wpe == 3 ? wpe : ((pl.PV == 6 && ntaps == 9 && !i3n_shape16(abf)) ? 3 : 2);
  return pl;
}

template <int NWN, int STATS, int PV, int WPE, int ABF = 0>
static int launch_i3n4(const Igemm3nParams& P, dim3 grid, size_t smem, hipStream_t s) {
  return P.d.ntaps == 9 ? launch_i3n6<NWN, 4 / NWN, STATS, PV, 1, WPE, ABF>(P, grid, smem, s)
                        : launch_i3n6<NWN, 4 / NWN, STATS, PV, 0, WPE, ABF>(P, grid, smem, s);
}

template <int NWN, int STATS>
static int launch_i3n(const Igemm3nParams& P, dim3 grid, size_t smem, hipStream_t s) {
  const I3nPlan pl = i3n_plan(P.d.NP, P.PP, P.d.ntaps, P.d.act_bf16);
  if (P.d.act_bf16) {
    if (pl.PV == 3) return launch_i3n4<NWN, STATS, 3, 3, 1>(P, grid, smem, s);
    if (pl.PV == 5) return launch_i3n4<NWN, STATS, 5, 3, 1>(P, grid, smem, s);
  } else {
    if (NWN == 4 && pl.PV == 6 && pl.WPE == 3) return launch_i3n4<NWN, STATS, 6, 3>(P, grid, smem, s);
    if (NWN == 4 && pl.PV == 9 && pl.WPE == 3) return launch_i3n4<NWN, STATS, 9, 3>(P, grid, smem, s);
    if (NWN == 4 && pl.PV == 6) return launch_i3n4<NWN, STATS, 6, 2>(P, grid, smem, s);
    if (NWN == 4 && pl.PV == 9) return launch_i3n4<NWN, STATS, 9, 2>(P, grid, smem, s);
    if (NWN != 4 && pl.PV == 6) return launch_i3n4<NWN, STATS, 6, 3>(P, grid, smem, s);
    if (NWN != 4 && pl.PV == 9) return launch_i3n4<NWN, STATS, 9, 3>(P, grid, smem, s);
  }
  bsed_set_error("bsed_igemm3n: patch of %d positions exceeds the 288 this build stages", P.PP);
  return BSED_ERR_ARG;
}

// NWN | MW << 4 | PV << 8 | WPE << 12 | ABF << 16 of the build bsed_igemm3n would launch (kernel labels of bench.py)
extern "C" int bsed_igemm3n_variant(const BsedIgemmDesc* d) {
  if (!d || d->TH <= 0 || d->TW <= 0) return -1;
  const int PP = (d->TW + 2 * d->hw) * (d->TH + 2 * d->hh);
  const I3nPlan pl = i3n_plan(d->NP, PP, d->ntaps, d->act_bf16);
  // bit 17: the 16 x 16 x 32 MFMA form (nine-tap instances, template argument SH)
  return pl.NWN | pl.MW << 4 | pl.PV << 8 | pl.WPE << 12 | (d->act_bf16 ? 1 : 0) << 16 |
         (((d->ntaps == 9 || (pl.NWN == 4 && pl.WPE == 2)) && i3n_shape16(d->act_bf16)) ? 1 : 0) << 17;
}

extern "C" int bsed_igemm3n_stats_rows(const BsedIgemmDesc* d) {
  if (!d || d->TH <= 0 || d->TW <= 0) return -1;
  const I3nPlan pl = i3n_plan(d->NP, (d->TW + 2 * d->hw) * (d->TH + 2 * d->hh), d->ntaps, d->act_bf16);
  return d->NB * ceil_div(d->H, d->TH) * (d->W / d->TW) * pl.MW;
}

extern "C" int bsed_igemm3n(const BsedIgemmDesc* desc, void* stream) {
  BSED_CHECK_ARG(desc, "bsed_igemm3n: null descriptor");
  Igemm3nParams P;
  P.d = *desc;
  BsedIgemmDesc& d = P.d;
  BSED_CHECK_ARG(d.in && d.w && d.out, "bsed_igemm3n: null tensor");
  BSED_CHECK_ARG(d.epilogue == BSED_EPI_PLAIN || d.epilogue == BSED_EPI_STATS, "bsed_igemm3n: PLAIN / STATS epilogues only");
  BSED_CHECK_ARG(d.epilogue != BSED_EPI_STATS || d.stats, "bsed_igemm3n: STATS needs a stats buffer");
  BSED_CHECK_ARG(d.NB > 0 && d.H > 0 && d.W > 0 && d.CIN > 0 && d.CIN % 32 == 0 && d.N > 0, "bsed_igemm3n: CIN must be a multiple of 32");
  BSED_CHECK_ARG(d.TH * d.TW == I3N_M && d.W % d.TW == 0, "bsed_igemm3n: TH*TW must be 128 and TW divide W");
  P.lgTW = 0;
  while ((1 << P.lgTW) < d.TW) ++P.lgTW;
  BSED_CHECK_ARG((1 << P.lgTW) == d.TW, "bsed_igemm3n: TW must be a power of two");
  BSED_CHECK_ARG(d.ntaps >= 1 && d.ntaps <= 9, "bsed_igemm3n: ntaps must be in 1..9");
  for (int t = 0; t < d.ntaps; ++t)
    BSED_CHECK_ARG(abs(d.dh[t]) <= d.hh && abs(d.dw[t]) <= d.hw, "bsed_igemm3n: tap %d outside the halo", t);
  BSED_CHECK_ARG(d.in_pitch >= d.CIN && d.in_pitch % (d.act_bf16 ? 8 : 4) == 0 && d.out_pitch >= d.N, "bsed_igemm3n: bad pitch");
  BSED_CHECK_ARG(d.NP % 32 == 0 && d.NP >= d.N, "bsed_igemm3n: NP must be N rounded up to 32");
  const int BN = d.NP % 128 == 0 ? 128 : (d.NP % 64 == 0 ? 64 : 32);
  d.tilesH = ceil_div(d.H, d.TH);
  d.tilesW = d.W / d.TW;
  P.PW = d.TW + 2 * d.hw;
  P.PH = d.TH + 2 * d.hh;
  P.PP = P.PW * P.PH;
  P.b_off = (P.PP * (d.act_bf16 ? 40 : I3N_ROW) + 7) & ~7;
  P.prio = (i3n_knob() & 8) ? 0 : 1;
  P.pw_magic = ((1 << 20) + P.PW - 1) / P.PW;
  for (int pos = 0; pos < P.PP; ++pos)
    BSED_CHECK_ARG(((pos * P.pw_magic) >> 20) == pos / P.PW, "bsed_igemm3n: internal: magic division fails for PW=%d", P.PW);
  // two patch buffers when the layer has more than one chunk
  const size_t bytes = (size_t)P.b_off * sizeof(unsigned short) * (d.CIN > I3N_KC ? 2 : 1);
  BSED_CHECK_ARG(bytes <= 160 * 1024, "bsed_igemm3n: tile needs %zu B of LDS", bytes);
  const long ntiles = (long)d.NB * d.tilesH * d.tilesW;
  BSED_CHECK_ARG(ntiles < (1L << 31), "bsed_igemm3n: too many tiles");
  BSED_CHECK_ARG((size_t)d.H * d.W * d.in_pitch < (1ull << 31), "bsed_igemm3n: an image of 2^31 elements or more");
  dim3 grid((unsigned)ntiles, d.NP / BN);
  hipStream_t s = (hipStream_t)stream;
  const bool st = d.epilogue == BSED_EPI_STATS;
  if (BN == 128) return st ? launch_i3n<4, 1>(P, grid, bytes, s) : launch_i3n<4, 0>(P, grid, bytes, s);
  if (BN == 64) return st ? launch_i3n<2, 1>(P, grid, bytes, s) : launch_i3n<2, 0>(P, grid, bytes, s);
  return st ? launch_i3n<1, 1>(P, grid, bytes, s) : launch_i3n<1, 0>(P, grid, bytes, s);
}
