// Waveform -> STFT magnitude -> mel -> (noise) -> dB, on gfx950.
//
// Replaces, on the GPU, what the reference does offline on the CPU through librosa:
//   preprocess()                /root/reference/src/data/preprocess.py:18-45
//   AugmentGaussianNoise        /root/reference/src/data/Transforms.py:155-196
//   ApplyLog (amplitude_to_db)  /root/reference/src/data/Transforms.py:74-86
//   PadOrTrunc                  /root/reference/src/data/Transforms.py:89-139
//
// Kernel plan (HBM-bound stage: 1.28 MB wav in + 0.64 MB mel out per 10 s clip at 32 kHz):
//   stft_mel_kernel : ONE WAVE PER FRAME, no workgroup barrier anywhere in the frame loop.  A wave takes SM_FPW
//                     consecutive frames of one clip.  Per frame each lane loads 16 complex points (even/odd sample
//                     pairs, 512 contiguous bytes per load instruction; the 87.5 % overlap between consecutive frames
//                     is served by L1/L2, HBM sees every sample about once), multiplies by the window held in
//                     registers, and runs a 1024-point complex FFT as 16 x 16 x 4:
//                       radix-16 in registers (points l + 64 j of lane l) -> twiddle W1024^(l k1) (registers)
//                       -> ONE exchange through a wave-private LDS tile (pitch 68: conflict-free both ways)
//                       -> radix-16 in registers -> twiddle W64^(m k2') -> radix-4 across the 4 lanes of a quad (DPP).
//                     The real-FFT unpack reads (Z[k], Z[1024-k]) pairs from the same tile, which yields bins k and
//                     1024-k at once; the 1025 magnitudes alias the tile; the sparse Slaney filterbank (2016 nnz) is
//                     applied as <= 17 float4 steps per lane over two bands (l and 127-l, long with short), weights
//                     in LDS.  Ordering inside a wave comes from the LDS's in-order execution (compiler barriers only).
//                     Only (T,128) linear mel goes to HBM; per-(clip, band) sums of squares leave as per-workgroup
//                     partials that mel_sumsq_finish_kernel adds in fixed order (no float atomics).
//   mel_noise_kernel: x + N(0, std_bin) with Philox/Box-Muller (or injected unit noise).
//   mel_db_kernel   : 10*log10(max(1e-10, x^2)) clamped to (clip max - 80 dB), zero pad/trunc.
#include "bsed_common.h"
#include "../../include/bsed.h"
#include <math.h>
#include <stdlib.h>
#include <string.h>
#include <algorithm>
#include <vector>

#define NFFT 2048
#define NC 1024        // complex points
#ifndef SM_WAVES
#define SM_WAVES 4     // waves per workgroup (independent of each other until the final partial-sum hand-off)
#endif
#ifndef SM_FPW
#define SM_FPW 8       // consecutive frames per wave
#endif
#ifndef SM_WPE
#define SM_WPE 2       // waves per SIMD the register budget is held to
#endif
#define SM_PITCH 68    // exchange tile: row pitch in complex elements (16 rows x 64 + 4 pad)
#define SM_TILE_FLOATS (2 * 16 * SM_PITCH)   // 8704 B per wave: exchange tile / Z image (1024 + 12 pad) / 1028 magnitudes
#define SM_MAX_NIT 24  // float4 filterbank steps per lane (both bands)
#define MEL_THREADS (64 * SM_WAVES)

struct MelPlan {
  BsedMelCfg cfg;
  int n_bins;
  float* d_window;   // [NFFT]
  float2* d_w1024;   // [1024]  e^{-2 pi i k/1024}
  float2* d_w2048;   // [1025]  e^{-2 pi i k/2048}
  int* d_mel_start;  // [n_mels]
  int* d_mel_count;  // [n_mels]
  int* d_mel_off;    // [n_mels]
  float* d_mel_w;    // [nnz]
  int nnz;
  // stft_mel_kernel tables
  float2* d_tw1;     // [15][64]  W1024^(lane * k1), k1 = 1..15
  float2* d_tw2;     // [15][64]  W64^((lane & 3) * k2), k2 = 1..15
  float2* d_wl;      // [64]      W2048^lane
  float4* d_melw4;   // [nit][64] filterbank weights of lane's bands (lane, n_mels-1-lane), zero padded
  int4* d_melidx;    // [64]      {s0, n0, s1, n1}: 4-aligned first bin and float4 count of the two bands
  int nit, nit0, nit1;   // float4 steps: nit0 for the bands 'lane', nit1 for the bands 'n_mels-1-lane', nit = nit0 + nit1
  bool pair_ok;          // stft_mel2_kernel's padded steps stay inside its tile (2 * S2_TILE_ELEMS frame pairs)
};

// ---------------------------------------------------------------------------------------------
// host: Slaney mel scale (librosa.filters.mel(htk=False, norm=None) semantics), double precision
// ---------------------------------------------------------------------------------------------
static double hz_to_mel(double f) {
  const double f_sp = 200.0 / 3.0, min_log_hz = 1000.0, min_log_mel = min_log_hz / f_sp;
  const double logstep = log(6.4) / 27.0;
  return f >= min_log_hz ? min_log_mel + log(f / min_log_hz) / logstep : f / f_sp;
}
static double mel_to_hz(double m) {
  const double f_sp = 200.0 / 3.0, min_log_hz = 1000.0, min_log_mel = min_log_hz / f_sp;
  const double logstep = log(6.4) / 27.0;
  return m >= min_log_mel ? min_log_hz * exp(logstep * (m - min_log_mel)) : f_sp * m;
}

static void build_filterbank(const BsedMelCfg& c, std::vector<int>& start, std::vector<int>& count,
                             std::vector<int>& off, std::vector<float>& w) {
  const int n_bins = c.n_fft / 2 + 1, nm = c.n_mels;
  std::vector<double> mel_f(nm + 2);
  const double lo = hz_to_mel(c.fmin), hi = hz_to_mel(c.fmax);
  for (int i = 0; i < nm + 2; ++i) mel_f[i] = mel_to_hz(lo + (hi - lo) * i / (nm + 1));
  start.assign(nm, 0); count.assign(nm, 0); off.assign(nm, 0); w.clear();
  for (int m = 0; m < nm; ++m) {
    const double d0 = mel_f[m + 1] - mel_f[m], d1 = mel_f[m + 2] - mel_f[m + 1];
    int first = -1, last = -1;
    std::vector<float> row(n_bins);
    for (int k = 0; k < n_bins; ++k) {
      const double f = (double)c.sr / 2.0 * k / (n_bins - 1);
      const double lower = (f - mel_f[m]) / d0, upper = (mel_f[m + 2] - f) / d1;
      const double v = fmax(0.0, fmin(lower, upper));
      row[k] = (float)v;
      if (row[k] > 0.f) { if (first < 0) first = k; last = k; }
    }
    off[m] = (int)w.size();
    if (first >= 0) {
      start[m] = first; count[m] = last - first + 1;
      for (int k = first; k <= last; ++k) w.push_back(row[k]);
    }
  }
}

// ---------------------------------------------------------------------------------------------
// device
// ---------------------------------------------------------------------------------------------
typedef float c32 __attribute__((ext_vector_type(2)));   // complex: .x re, .y im

__device__ __forceinline__ c32 cmulw(c32 a, c32 w) { return c32{a.x * w.x - a.y * w.y, a.x * w.y + a.y * w.x}; }
__device__ __forceinline__ c32 mul_mi(c32 a) { return c32{a.y, -a.x}; }   // * (-i)

// 16-point DFT, natural order in and out, as 4 x 4: v[4a+b] -> X[c+4d] with W16^(b c) between the two radix-4 levels
__device__ __forceinline__ void dft16(c32 (&v)[16]) {
  const float C1 = 0.92387953251128674f, S1 = 0.38268343236508977f, R = 0.70710678118654752f;
  c32 t[4][4];
#pragma unroll
  for (int b = 0; b < 4; ++b) {
    const c32 s02 = v[b] + v[8 + b], d02 = v[b] - v[8 + b];
    const c32 s13 = v[4 + b] + v[12 + b], d13 = mul_mi(v[4 + b] - v[12 + b]);
    t[b][0] = s02 + s13; t[b][1] = d02 + d13; t[b][2] = s02 - s13; t[b][3] = d02 - d13;
  }
  t[1][1] = cmulw(t[1][1], c32{C1, -S1});                                       // W16^1
  t[1][2] = c32{(t[1][2].x + t[1][2].y) * R, (t[1][2].y - t[1][2].x) * R};      // W16^2 = (1 - i)/sqrt2
  t[1][3] = cmulw(t[1][3], c32{S1, -C1});                                       // W16^3
  t[2][1] = c32{(t[2][1].x + t[2][1].y) * R, (t[2][1].y - t[2][1].x) * R};      // W16^2
  t[2][2] = mul_mi(t[2][2]);                                                    // W16^4 = -i
  t[2][3] = c32{(t[2][3].y - t[2][3].x) * R, -(t[2][3].x + t[2][3].y) * R};     // W16^6 = (-1 - i)/sqrt2
  t[3][1] = cmulw(t[3][1], c32{S1, -C1});                                       // W16^3
  t[3][2] = c32{(t[3][2].y - t[3][2].x) * R, -(t[3][2].x + t[3][2].y) * R};     // W16^6
  t[3][3] = cmulw(t[3][3], c32{-C1, S1});                                       // W16^9
#pragma unroll
  for (int c = 0; c < 4; ++c) {
    const c32 s02 = t[0][c] + t[2][c], d02 = t[0][c] - t[2][c];
    const c32 s13 = t[1][c] + t[3][c], d13 = mul_mi(t[1][c] - t[3][c]);
    v[c] = s02 + s13; v[c + 4] = d02 + d13; v[c + 8] = s02 - s13; v[c + 12] = d02 - d13;
  }
}

template <int CTRL>
__device__ __forceinline__ float quad_dpp(float x) {   // lane exchange inside a quad (v_mov_b32_dpp quad_perm)
  return __int_as_float(__builtin_amdgcn_mov_dpp(__float_as_int(x), CTRL, 0xf, 0xf, true));
}

#define SM_FENCE() do { asm volatile("" ::: "memory"); __builtin_amdgcn_sched_barrier(0); } while (0)   // LDS executes a wave's accesses in order: only the compiler must not reorder

struct SmParams {
  const float* wav; int n_samples, hop, T, n_mels;
  const float* window; const float2* tw1; const float2* tw2; const float2* wl;
  const float4* melw4; const int4* melidx; int nit, nit0, nit1;
  float* mel_out; float* clip_max; float* sumsq_part;   // sumsq_part (B, gridDim.x, n_mels)
};

__global__ __launch_bounds__(MEL_THREADS, SM_WPE) void stft_mel_kernel(const SmParams P) {
  extern __shared__ __align__(16) unsigned char smem_raw[];
  float4* melw = reinterpret_cast<float4*>(smem_raw);                                     // [nit][64]
  c32* wins = reinterpret_cast<c32*>(smem_raw + (size_t)P.nit * 64 * sizeof(float4));     // [1024] window, (even, odd) pairs
  float* tiles = reinterpret_cast<float*>(wins + NC);                                     // [SM_WAVES][SM_TILE_FLOATS]
  const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
  const int b = blockIdx.y;
  float* tile = tiles + wv * SM_TILE_FLOATS;
  c32* xb = reinterpret_cast<c32*>(tile);

  for (int i = tid; i < P.nit * 64; i += MEL_THREADS) melw[i] = P.melw4[i];
  for (int i = tid; i < NC; i += MEL_THREADS) {
    const float2 ww = *reinterpret_cast<const float2*>(P.window + 2 * i);
    wins[i] = c32{ww.x, ww.y};
  }
  // per-lane constants (registers for the whole kernel).  Twiddle k = 4a + b is held as the two factors
  // W^(lane 4a) and W^(lane b) (6 complex numbers instead of 15 per pass; one extra multiply for a, b != 0)
#if SM_WPE <= 2
  // 256-register budget: all 15 + 15 twiddles of the two passes stay in registers
  c32 tw1[15], tw2[15];
#pragma unroll
  for (int k = 0; k < 15; ++k) {
    const float2 a = P.tw1[k * 64 + lane], c = P.tw2[k * 64 + lane];
    tw1[k] = c32{a.x, a.y}; tw2[k] = c32{c.x, c.y};
  }
#else
  c32 t1a[3], t1b[3], t2a[3], t2b[3];
#pragma unroll
  for (int i = 0; i < 3; ++i) {
    const float2 a1 = P.tw1[(4 * (i + 1) - 1) * 64 + lane], b1 = P.tw1[i * 64 + lane];
    const float2 a2 = P.tw2[(4 * (i + 1) - 1) * 64 + lane], b2 = P.tw2[i * 64 + lane];
    t1a[i] = c32{a1.x, a1.y}; t1b[i] = c32{b1.x, b1.y}; t2a[i] = c32{a2.x, a2.y}; t2b[i] = c32{b2.x, b2.y};
  }
#endif
  const float2 wl2 = P.wl[lane];
  const c32 wl = c32{wl2.x, wl2.y};
  const int4 mi = P.melidx[lane];
  const int g = lane >> 2, m = lane & 3;
  const float sgn2 = (m & 2) ? -1.f : 1.f, sgn1 = (m & 1) ? -1.f : 1.f;
  const int q = ((m & 1) << 1) | (m >> 1);          // which of the quad's four outputs this lane ends up with
  const int band0 = lane, band1 = P.n_mels - 1 - lane;
  const bool has0 = band0 < P.n_mels && band0 <= band1, has1 = band1 >= 0 && band1 > band0;
  __syncthreads();   // filterbank table staged (the only barrier before the final partial-sum hand-off)

  const float* w = P.wav + (size_t)b * P.n_samples;
  const int t_begin = (blockIdx.x * SM_WAVES + wv) * SM_FPW;
  float run_max = 0.f, sq0 = 0.f, sq1 = 0.f;
  c32 v[16];
  int opq = 0;   // opaque zero, re-laundered per frame: keeps loop-invariant loads / products out of long-lived registers
  // interior frames: 16 loads of 8 bytes straight into the butterfly registers
  auto interior = [&](int t) {
    const long s0 = (long)t * P.hop - NFFT / 2;
    return s0 >= 0 && s0 + NFFT <= (long)P.n_samples;
  };
  auto load_fast = [&](int t) {
    const float* src = w + ((long)t * P.hop - NFFT / 2) + 2 * lane;   // 4-byte aligned pairs: dwordx2 at dword alignment
#pragma unroll
    for (int j = 0; j < 16; ++j) {
      float2 x;
      __builtin_memcpy(&x, src + 128 * j, sizeof(x));
      v[j] = c32{x.x, x.y};
    }
  };
  // clip edges (librosa.stft(center=True, pad_mode='reflect')): the ~10 edge frames of a clip are staged through the
  // wave's tile by a rolled loop (a second unrolled copy of the loads with mirrored indices cost 24 registers for the
  // whole kernel), then picked up with the same register mapping
  auto load_edge = [&](int t) {
    const long s0 = (long)t * P.hop - NFFT / 2;
#pragma unroll 1
    for (int i = 0; i < NFFT / 64; ++i) {
      long gi = s0 + 64 * i + lane;
      if (gi < 0) gi = -gi;
      if (gi >= P.n_samples) gi = 2L * (P.n_samples - 1) - gi;
      gi = gi < 0 ? 0 : (gi >= P.n_samples ? P.n_samples - 1 : gi);
      tile[64 * i + lane] = w[gi];
    }
    SM_FENCE();
#pragma unroll
    for (int j = 0; j < 16; ++j) v[j] = xb[lane + 64 * j];
    SM_FENCE();
  };
  if (t_begin < P.T) {
    if (interior(t_begin)) load_fast(t_begin); else load_edge(t_begin);
  }
#pragma unroll 1
  for (int f = 0; f < SM_FPW; ++f) {
    const int t = t_begin + f;
    if (t >= P.T) break;                             // wave-uniform
    asm volatile("" : "+v"(opq));
#if SM_WPE <= 2
    const c32 wlo = wl;                              // the eight W2048^k of the unpack are hoisted into registers too
#else
#pragma unroll
    for (int i = 0; i < 3; ++i) {                    // the nine twiddle products below are recomputed per frame, not hoisted
      asm volatile("" : "+v"(t1a[i].x), "+v"(t1a[i].y));
      asm volatile("" : "+v"(t2a[i].x), "+v"(t2a[i].y));
    }
    c32 wlo = wl;                                    // likewise the eight W2048^k of the unpack
    asm volatile("" : "+v"(wlo.x), "+v"(wlo.y));
#endif
#pragma unroll
    for (int j = 0; j < 16; ++j) v[j] *= wins[opq + lane + 64 * j];   // window from LDS (read per frame: 32 registers saved)
    // ---- pass 1: 16-point DFT over j of z[l + 64 j], twiddle W1024^(l k1)
    dft16(v);
#pragma unroll
    for (int k = 1; k < 16; ++k) {
#if SM_WPE <= 2
      v[k] = cmulw(v[k], tw1[k - 1]);
#else
      const int a = k >> 2, bb = k & 3;
      const c32 tw = a == 0 ? t1b[bb - 1] : (bb == 0 ? t1a[a - 1] : cmulw(t1a[a - 1], t1b[bb - 1]));
      v[k] = cmulw(v[k], tw);
#endif
    }
    SM_FENCE();
#pragma unroll
    for (int k = 0; k < 16; ++k) xb[k * SM_PITCH + lane] = v[k];
    SM_FENCE();
    // ---- pass 2: lane (g = k1, m) takes points l = m + 4 j' of row k1
#pragma unroll
    for (int j = 0; j < 16; ++j) v[j] = xb[g * SM_PITCH + m + 4 * j];
    SM_FENCE();
    dft16(v);
#pragma unroll
    for (int k = 1; k < 16; ++k) {
#if SM_WPE <= 2
      v[k] = cmulw(v[k], tw2[k - 1]);
#else
      const int a = k >> 2, bb = k & 3;
      const c32 tw = a == 0 ? t2b[bb - 1] : (bb == 0 ? t2a[a - 1] : cmulw(t2a[a - 1], t2b[bb - 1]));
      v[k] = cmulw(v[k], tw);
#endif
    }
    // ---- pass 3: 4-point DFT across the quad's lanes; lane m ends with output q(m): k = g + 16 k2' + 256 q
#pragma unroll
    for (int k = 0; k < 16; ++k) {
      c32 o = c32{quad_dpp<0x4E>(v[k].x), quad_dpp<0x4E>(v[k].y)};                 // lane ^ 2
      c32 r = c32{fmaf(v[k].x, sgn2, o.x), fmaf(v[k].y, sgn2, o.y)};                // m<2: D + other, m>=2: other - D
      r = (m == 3) ? mul_mi(r) : r;
      o = c32{quad_dpp<0xB1>(r.x), quad_dpp<0xB1>(r.y)};                            // lane ^ 1
      v[k] = c32{fmaf(r.x, sgn1, o.x), fmaf(r.y, sgn1, o.y)};
    }
    // ---- Z image: element k at k + 4 (k >> 8) (conflict-free 8-byte writes from the (g, q) lane pattern)
#pragma unroll
    for (int k = 0; k < 16; ++k) xb[g + 16 * k + 260 * q] = v[k];
    // The partner of bin k = lane + 64 i sits at 1024 - k + 4 ((1024 - k) >> 8) = 1036 - lane - 64 i - 4 (i >> 2) for every
    // (lane, i) except lane 0 with i = 0 (partner Z[0]) and i = 4 (partner Z[768]): those two get a copy in the pad
    // holes the formula points at, so that every pair address is one per-lane base plus an immediate
    if (lane == 0) xb[1036] = v[0];
    if (lane == 3) xb[776] = v[0];
    SM_FENCE();
    // ---- real-FFT unpack on pairs (k, 1024 - k), k = lane + 64 i: |E + t|, |E - t| are bins k and 1024 - k
    float mg_lo[8], mg_hi[8];
    const c32* xb_rev = xb + (1036 - (64 * 7 + 4)) - lane;   // + positive immediates only
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      const c32 zk = xb[lane + 64 * i + 4 * (i >> 2)];
      const c32 zr = xb_rev[(64 * 7 + 4) - (64 * i + 4 * (i >> 2))];
      const c32 e = c32{zk.x + zr.x, zk.y - zr.y};                 // 2 E
      const c32 o = c32{zk.y + zr.y, zr.x - zk.x};                 // 2 O = (Zk - conj Zr) / i
      const float a = 6.283185307179586477f * (float)(64 * i) / (float)NFFT;
      const c32 wk = cmulw(wlo, c32{__builtin_cosf(a), -__builtin_sinf(a)});   // W2048^k = W2048^lane * W32^i (constant folded)
      const c32 tt = cmulw(o, wk);
      const c32 xp = e + tt, xm = e - tt;
      mg_lo[i] = 0.5f * __builtin_amdgcn_sqrtf(xp.x * xp.x + xp.y * xp.y);   // v_sqrt_f32 (1 ulp)
      mg_hi[i] = 0.5f * __builtin_amdgcn_sqrtf(xm.x * xm.x + xm.y * xm.y);
    }
    const c32 z512 = xb[512 + 4 * 2];
    const float mg512 = __builtin_amdgcn_sqrtf(z512.x * z512.x + z512.y * z512.y);
    SM_FENCE();
    // ---- magnitudes over the tile (1025 + 3 zero pad so that float4 filterbank steps may run past bin 1024)
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      tile[lane + 64 * i] = mg_lo[i];
      (tile + (NC - 64 * 7) - lane)[64 * (7 - i)] = mg_hi[i];
    }
    if (lane == 0) tile[512] = mg512;
    if (lane >= 1 && lane <= 3) tile[NC + lane] = 0.f;
    SM_FENCE();
    // next frame's samples travel while the filterbank runs
    const bool more = f + 1 < SM_FPW && t + 1 < P.T;
    const bool fast = more && interior(t + 1);
    if (fast) load_fast(t + 1);
    // ---- sparse filterbank: two bands per lane, float4 steps
    // two uniform loops (zero-weight steps pad the shorter bands; their magnitude reads stay inside the tile, whose
    // floats are all finite here: bins, zero pad, and older Z words)
    float acc0 = 0.f, acc1 = 0.f;
    const float* m0 = tile + mi.x;
    const float* m1 = tile + mi.z;
    const float4* wq = melw + lane;
    for (int it = 0; it < P.nit0; ++it) {
      const float4 mg = *reinterpret_cast<const float4*>(m0 + 4 * it);
      const float4 ww = wq[it * 64];
      acc0 = fmaf(ww.x, mg.x, fmaf(ww.y, mg.y, fmaf(ww.z, mg.z, fmaf(ww.w, mg.w, acc0))));
    }
    wq += P.nit0 * 64;
    for (int it = 0; it < P.nit1; ++it) {
      const float4 mg = *reinterpret_cast<const float4*>(m1 + 4 * it);
      const float4 ww = wq[it * 64];
      acc1 = fmaf(ww.x, mg.x, fmaf(ww.y, mg.y, fmaf(ww.z, mg.z, fmaf(ww.w, mg.w, acc1))));
    }
    SM_FENCE();
    float* out = P.mel_out + ((size_t)b * P.T + t) * P.n_mels;
    if (has0) { out[band0] = acc0; run_max = fmaxf(run_max, acc0); sq0 = fmaf(acc0, acc0, sq0); }
    if (has1) { out[band1] = acc1; run_max = fmaxf(run_max, acc1); sq1 = fmaf(acc1, acc1, sq1); }
    if (more && !fast) load_edge(t + 1);             // the tile is free again (magnitudes consumed)
  }
  const float wm = wave_max(run_max);
  if (lane == 0 && t_begin < P.T) atomicMax(reinterpret_cast<int*>(P.clip_max + b), __float_as_int(wm));
  // per-workgroup partial sums of squares, waves added in fixed order
  SM_FENCE();
  tile[lane] = sq0; tile[64 + lane] = sq1;
  __syncthreads();
  if (wv == 0) {
    float s0 = 0.f, s1 = 0.f;
#pragma unroll
    for (int u = 0; u < SM_WAVES; ++u) { s0 += tiles[u * SM_TILE_FLOATS + lane]; s1 += tiles[u * SM_TILE_FLOATS + 64 + lane]; }
    float* part = P.sumsq_part + ((size_t)b * gridDim.x + blockIdx.x) * P.n_mels;
    if (has0) part[band0] = s0;
    if (has1) part[band1] = s1;
  }
}

// ---------------------------------------------------------------------------------------------
// stft_mel2_kernel: TWO FRAMES PER WAVE (the default; stft_mel_kernel above remains for filterbanks whose tables do not
// fit beside eight pair tiles, and for A/B runs with BSED_MEL_PAIR=0).  Every quantity is held as the PAIR
// (frame t, frame t+1) in an even-aligned register pair, so that every add / multiply / FMA of the FFT, the unpack and
// the filterbank is one v_pk_*_f32 over both frames, with the lane's twiddle / window / filter weight as a broadcast
// operand (op_sel; no swizzle of a computed value anywhere).  The samples arrive as (even, odd) pairs per frame: the
// window multiplies them in that layout and ONE v_swap_b32 per point turns (e_t, o_t), (e_t+1, o_t+1) into
// (e_t, e_t+1), (o_t, o_t+1).  Same decomposition (16 x 16 x 4) and the same operation order per frame as
// stft_mel_kernel (the two agree to a few ulp: the compiler picks the fused multiply-adds independently).
// LDS elements are float4 {re_t, re_t+1, im_t, im_t+1}; a wave's tile is 1088 of them (17 KB: exchange rows of pitch 68,
// Z image at k + 4 (k >> 8), then the (t, t+1) magnitudes); eight waves per workgroup, one workgroup per CU: 136 KB of
// tiles + the filterbank table + HALF the window (it is symmetric, w[n] = w[2047 - n]) + the quad twiddles.
// Measured (MI355X, 256 clips x 865 frames): 612 VALU instructions per frame against 852, 0.55 ms against 0.60 ms.
// Neither kernel is bound by one pipe: VALU 48 % / LDS 49 % busy (a third of it bank conflicts of the filterbank's
// gather) / texture addresser 28 %, two waves per SIMD -- DESIGN.md section 5.
// ---------------------------------------------------------------------------------------------
#define S2_WAVES 8
#define S2_THREADS (64 * S2_WAVES)
#ifndef S2_FPW
#define S2_FPW 16             // consecutive frames per wave (pairs: S2_FPW / 2)
#endif
#define S2_TILE_ELEMS 1088    // float4 elements per wave (16 exchange rows of pitch 68)
#ifndef S2_FB_UNROLL
#define S2_FB_UNROLL 4
#endif
#ifndef S2_TIMING
#define S2_TIMING 0     // 1: one wave prints the cycles it spent in each phase of a frame pair (tools/mel_pair_ab.py prof)
#endif
#if S2_TIMING
#define S2_STAMP(i) do { asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); const unsigned long long now_ = __builtin_amdgcn_s_memtime(); \
                         asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); ph[i] += (unsigned)(now_ - last_); last_ = now_; } while (0)
#else
#define S2_STAMP(i) do {} while (0)
#endif
#ifndef S2_EARLY_PREFETCH
#define S2_EARLY_PREFETCH 1
#endif
#ifndef S2_STAGGER
#define S2_STAGGER 0
#endif
#ifndef S2_DFT_SB
#define S2_DFT_SB 1
#endif
#ifndef S2_TW1_FACTORED
#define S2_TW1_FACTORED 1
#endif
#define S2_ROW(k) ((k) * 68)

typedef float p2 __attribute__((ext_vector_type(2)));   // one quantity of frames (t, t+1)
typedef float f4 __attribute__((ext_vector_type(4)));
struct cp { p2 re, im; };

__device__ __forceinline__ p2 bc(float s) { return p2{s, s}; }
__device__ __forceinline__ cp operator+(cp a, cp b) { return cp{a.re + b.re, a.im + b.im}; }
__device__ __forceinline__ cp operator-(cp a, cp b) { return cp{a.re - b.re, a.im - b.im}; }
__device__ __forceinline__ cp cmulw(cp a, c32 w) {
  const p2 wx = bc(w.x), wy = bc(w.y);
  return cp{a.re * wx - a.im * wy, a.re * wy + a.im * wx};
}
__device__ __forceinline__ cp mul_mi(cp a) { return cp{a.im, -a.re}; }                        // * (-i)
__device__ __forceinline__ cp mul_w8(cp a, p2 R) { return cp{(a.re + a.im) * R, (a.im - a.re) * R}; }     // * (1 - i)/sqrt2
__device__ __forceinline__ cp mul_w8_3(cp a, p2 R) { return cp{(a.im - a.re) * R, -((a.re + a.im) * R)}; }   // * (-1 - i)/sqrt2
__device__ __forceinline__ f4 pack4(cp a) { return f4{a.re.x, a.re.y, a.im.x, a.im.y}; }
__device__ __forceinline__ cp unpack4(f4 a) { return cp{p2{a.x, a.y}, p2{a.z, a.w}}; }

#if S2_DFT_SB
#define S2_SB() __builtin_amdgcn_sched_barrier(0)   // one radix-4 column at a time: bounds the live temporaries
#else
#define S2_SB() do {} while (0)
#endif
// dft16 of stft_mel_kernel on frame pairs (same operation order per frame)
__device__ __forceinline__ void dft16(cp (&v)[16]) {
  const float C1 = 0.92387953251128674f, S1 = 0.38268343236508977f;
  const p2 R = bc(0.70710678118654752f);
  cp t[4][4];
#pragma unroll
  for (int b = 0; b < 4; ++b) {
    const cp s02 = v[b] + v[8 + b], d02 = v[b] - v[8 + b];
    const cp s13 = v[4 + b] + v[12 + b], d13 = mul_mi(v[4 + b] - v[12 + b]);
    t[b][0] = s02 + s13; t[b][1] = d02 + d13; t[b][2] = s02 - s13; t[b][3] = d02 - d13;
    S2_SB();
  }
  t[1][1] = cmulw(t[1][1], c32{C1, -S1});
  t[1][2] = mul_w8(t[1][2], R);
  t[1][3] = cmulw(t[1][3], c32{S1, -C1});
  t[2][1] = mul_w8(t[2][1], R);
  t[2][2] = mul_mi(t[2][2]);
  t[2][3] = mul_w8_3(t[2][3], R);
  t[3][1] = cmulw(t[3][1], c32{S1, -C1});
  t[3][2] = mul_w8_3(t[3][2], R);
  t[3][3] = cmulw(t[3][3], c32{-C1, S1});
  S2_SB();
#pragma unroll
  for (int c = 0; c < 4; ++c) {
    const cp s02 = t[0][c] + t[2][c], d02 = t[0][c] - t[2][c];
    const cp s13 = t[1][c] + t[3][c], d13 = mul_mi(t[1][c] - t[3][c]);
    v[c] = s02 + s13; v[c + 4] = d02 + d13; v[c + 8] = s02 - s13; v[c + 12] = d02 - d13;
    S2_SB();
  }
}

template <int CTRL>
__device__ __forceinline__ p2 quad_dpp2(p2 x) { return p2{quad_dpp<CTRL>(x.x), quad_dpp<CTRL>(x.y)}; }

__global__ __launch_bounds__(S2_THREADS, 2) void stft_mel2_kernel(const SmParams P) {
  extern __shared__ __align__(16) unsigned char smem_raw[];
  float4* melw = reinterpret_cast<float4*>(smem_raw);                                     // [nit][64]
  c32* wins = reinterpret_cast<c32*>(smem_raw + (size_t)P.nit * 64 * sizeof(float4));     // [512] first half of the window, (even, odd) pairs
  c32* tw2s = wins + NC / 2;                                                              // [15][4] (+ 4 pad)
  f4* tiles = reinterpret_cast<f4*>(tw2s + 64);                                           // [S2_WAVES][S2_TILE_ELEMS]
  const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
  const int b = blockIdx.y;
  f4* xt = tiles + wv * S2_TILE_ELEMS;
  float* tile = reinterpret_cast<float*>(xt);
  p2* mg2 = reinterpret_cast<p2*>(xt);

  for (int i = tid; i < P.nit * 64; i += S2_THREADS) melw[i] = P.melw4[i];
  for (int i = tid; i < NC / 2; i += S2_THREADS) {
    const float2 ww = *reinterpret_cast<const float2*>(P.window + 2 * i);
    wins[i] = c32{ww.x, ww.y};
  }
  // padded filterbank steps (zero weights) may read any word of the tile: none may hold a NaN pattern
  for (int i = lane; i < S2_TILE_ELEMS; i += 64) xt[i] = f4{0.f, 0.f, 0.f, 0.f};
  // W1024^(lane k) in registers; W64^((lane & 3) k) depends on the lane's place in its quad only: a 15 x 4 LDS table
#if S2_TW1_FACTORED
  // twiddle k = 4 a + b as the two factors W^(lane 4 a), W^(lane b): 6 complex numbers instead of 15 (the nine products
  // are recomputed per pair: 36 scalar operations against 24 registers)
  c32 t1a[3], t1b[3];
#pragma unroll
  for (int i = 0; i < 3; ++i) {
    const float2 a1 = P.tw1[(4 * (i + 1) - 1) * 64 + lane], b1 = P.tw1[i * 64 + lane];
    t1a[i] = c32{a1.x, a1.y}; t1b[i] = c32{b1.x, b1.y};
  }
#else
  c32 tw1[15];
#pragma unroll
  for (int k = 0; k < 15; ++k) {
    const float2 a = P.tw1[k * 64 + lane];
    tw1[k] = c32{a.x, a.y};
  }
#endif
  if (tid < 60) {
    const float2 c = P.tw2[(tid >> 2) * 64 + (tid & 3)];
    tw2s[tid] = c32{c.x, c.y};
  }
  const float2 wl2 = P.wl[lane];
  const c32 wl = c32{wl2.x, wl2.y};
  const int4 mi = P.melidx[lane];
  const int g = lane >> 2, m = lane & 3;
  const p2 sgn2 = bc((m & 2) ? -1.f : 1.f), sgn1 = bc((m & 1) ? -1.f : 1.f);
  const bool m3 = m == 3;
  const int q = ((m & 1) << 1) | (m >> 1);
  const int band0 = lane, band1 = P.n_mels - 1 - lane;
  const bool has0 = band0 < P.n_mels && band0 <= band1, has1 = band1 >= 0 && band1 > band0;
  __syncthreads();

  const float* w = P.wav + (size_t)b * P.n_samples;
  const int t_begin = (blockIdx.x * S2_WAVES + wv) * S2_FPW;
  float run_max = 0.f;
  p2 sq0 = bc(0.f), sq1 = bc(0.f);
  // xa[j] / xb[j]: raw (even, odd) samples of frame t / t+1 as loaded; after the window and one register swap per
  // point the same registers hold (re_t, re_t+1) / (im_t, im_t+1)
  p2 xa[16], xb[16];
  int opq = 0;
  auto interior = [&](int t) {
    const long s0 = (long)t * P.hop - NFFT / 2;
    return s0 >= 0 && s0 + NFFT <= (long)P.n_samples;
  };
  auto load_fast = [&](int ta, int tb) {
    const float* sa = w + ((long)ta * P.hop - NFFT / 2) + 2 * lane;
    const float* sb = w + ((long)tb * P.hop - NFFT / 2) + 2 * lane;
#pragma unroll
    for (int j = 0; j < 16; ++j) {
      __builtin_memcpy(&xa[j], sa + 128 * j, sizeof(float2));
      __builtin_memcpy(&xb[j], sb + 128 * j, sizeof(float2));
    }
  };
  // clip edges (reflect padding): both frames of the pair are staged through the tile (frame t in floats [0, 2048),
  // frame t+1 in [2048, 4096)) by a rolled loop, then picked up with the register mapping of the fast path
  auto load_edge = [&](int ta, int tb) {
#pragma unroll 1
    for (int h = 0; h < 2; ++h) {
      const long s0 = (long)(h ? tb : ta) * P.hop - NFFT / 2;
#pragma unroll 1
      for (int i = 0; i < NFFT / 64; ++i) {
        long gi = s0 + 64 * i + lane;
        if (gi < 0) gi = -gi;
        if (gi >= P.n_samples) gi = 2L * (P.n_samples - 1) - gi;
        gi = gi < 0 ? 0 : (gi >= P.n_samples ? P.n_samples - 1 : gi);
        tile[h * NFFT + 64 * i + lane] = w[gi];
      }
    }
    SM_FENCE();
    const p2* ea = reinterpret_cast<const p2*>(tile);
#pragma unroll
    for (int j = 0; j < 16; ++j) { xa[j] = ea[lane + 64 * j]; xb[j] = ea[NC + lane + 64 * j]; }
    SM_FENCE();
  };
  auto load_pair = [&](int ta) {   // wave-uniform choice
    const int tb = ta + 1 < P.T ? ta + 1 : ta;   // a clip's odd last frame is paired with itself (second result dropped)
    if (interior(ta) && interior(tb)) load_fast(ta, tb); else load_edge(ta, tb);
  };
  if (t_begin < P.T) load_pair(t_begin);
#if S2_STAGGER
  // the eight waves of a workgroup start together and would run every phase (VALU / LDS / loads) in step, queueing on
  // one resource while the others idle: start wave w  w * S2_STAGGER cycles late
  for (int i = 0; i < wv * (S2_STAGGER / 64); ++i) __builtin_amdgcn_s_sleep(1);
#endif
#if S2_TIMING
  unsigned ph[10] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
  unsigned long long last_ = __builtin_amdgcn_s_memtime();
  const unsigned long long first_ = last_;
#endif
#pragma unroll 1
  for (int f = 0; f < S2_FPW; f += 2) {
    const int t = t_begin + f;
    if (t >= P.T) break;                             // wave-uniform
    const bool two = t + 1 < P.T;
    asm volatile("" : "+v"(opq));
    c32 wlo = wl;                                    // the eight W2048^k of the unpack are recomputed per pair, not hoisted
    asm volatile("" : "+v"(wlo.x), "+v"(wlo.y));
#if S2_TW1_FACTORED
#pragma unroll
    for (int i = 0; i < 3; ++i) asm volatile("" : "+v"(t1a[i].x), "+v"(t1a[i].y));   // likewise the nine twiddle products
#endif
    cp v[16];
#pragma unroll
    for (int j = 0; j < 16; ++j) {
      // samples 2 i, 2 i + 1 with i = lane + 64 j; for i >= 512 the window pair is the mirrored pair 1023 - i, swapped
      const c32 wm = j < 8 ? wins[opq + lane + 64 * j] : (wins + (NC / 2 - 1) - lane)[opq + 64 * (15 - j) - 64 * 7 - (NC / 2 - 64 * 8)];
      const c32 wn = j < 8 ? wm : c32{wm.y, wm.x};
      xa[j] *= wn; xb[j] *= wn;                                            // (even w_even, odd w_odd) of each frame
      asm("v_swap_b32 %0, %1" : "+v"(xa[j].y), "+v"(xb[j].x));             // -> (even_t, even_t+1), (odd_t, odd_t+1)
      v[j] = cp{xa[j], xb[j]};
    }
#if S2_TIMING
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#pragma unroll
    for (int j = 0; j < 16; ++j) asm volatile("" :: "v"(v[j].re), "v"(v[j].im));
#endif
    S2_STAMP(0);
    // ---- pass 1
    dft16(v);
#pragma unroll
    for (int k = 1; k < 16; ++k) {
#if S2_TW1_FACTORED
      const int a = k >> 2, bb = k & 3;
      const c32 tw = a == 0 ? t1b[bb - 1] : (bb == 0 ? t1a[a - 1] : cmulw(t1a[a - 1], t1b[bb - 1]));
      v[k] = cmulw(v[k], tw);
#else
      v[k] = cmulw(v[k], tw1[k - 1]);
#endif
    }
    SM_FENCE();
#if S2_TIMING
#pragma unroll
    for (int j = 0; j < 16; ++j) asm volatile("" :: "v"(v[j].re), "v"(v[j].im));
#endif
    S2_STAMP(1);
#pragma unroll
    for (int k = 0; k < 16; ++k) xt[S2_ROW(k) + lane] = pack4(v[k]);
    SM_FENCE();
    // ---- pass 2: lane (g = k1, m) takes points l = m + 4 j' of row k1
    {
      const f4* row = xt + (S2_ROW(g) + m);
#pragma unroll
      for (int j = 0; j < 16; ++j) v[j] = unpack4(row[4 * j]);
    }
    SM_FENCE();
    S2_STAMP(2);
    dft16(v);
#pragma unroll
    for (int k = 1; k < 16; ++k) v[k] = cmulw(v[k], tw2s[opq + 4 * (k - 1) + m]);
#if S2_TIMING
#pragma unroll
    for (int j = 0; j < 16; ++j) asm volatile("" :: "v"(v[j].re), "v"(v[j].im));
    __builtin_amdgcn_sched_barrier(0);
#endif
    S2_STAMP(3);
    // ---- pass 3: 4-point DFT across the quad's lanes
#pragma unroll
    for (int k = 0; k < 16; ++k) {
      cp o = cp{quad_dpp2<0x4E>(v[k].re), quad_dpp2<0x4E>(v[k].im)};               // lane ^ 2
      cp r = cp{v[k].re * sgn2 + o.re, v[k].im * sgn2 + o.im};
      const cp rr = mul_mi(r);
      r.re = m3 ? rr.re : r.re; r.im = m3 ? rr.im : r.im;
      o = cp{quad_dpp2<0xB1>(r.re), quad_dpp2<0xB1>(r.im)};                        // lane ^ 1
      v[k] = cp{r.re * sgn1 + o.re, r.im * sgn1 + o.im};
#ifdef S2_SB_DPP
      if ((k & (S2_SB_DPP - 1)) == S2_SB_DPP - 1) __builtin_amdgcn_sched_barrier(0);   // bounds the live temporaries of the stage
#endif
    }
#if S2_TIMING
#pragma unroll
    for (int j = 0; j < 16; ++j) asm volatile("" :: "v"(v[j].re), "v"(v[j].im));
    __builtin_amdgcn_sched_barrier(0);
#endif
    S2_STAMP(4);
    // ---- Z image: element k at k + 4 (k >> 8)
#pragma unroll
    for (int k = 0; k < 16; ++k) xt[g + 16 * k + 260 * q] = pack4(v[k]);
    if (lane == 0) xt[1036] = pack4(v[0]);
    if (lane == 3) xt[776] = pack4(v[0]);
    SM_FENCE();
    // the butterfly registers are free from here on: the next pair's samples travel during the unpack and the filterbank
    const bool more = f + 2 < S2_FPW && t + 2 < P.T;
    const int tn = t + 2, tnb = t + 3 < P.T ? t + 3 : t + 2;
    const bool fast = more && interior(tn) && interior(tnb);
#if S2_EARLY_PREFETCH
    if (fast) load_fast(tn, tnb);
#endif
    // ---- real-FFT unpack on pairs (k, 1024 - k), k = lane + 64 i
    p2 mg_lo[8], mg_hi[8];
    const f4* xt_rev = xt + (1036 - (64 * 7 + 4)) - lane;
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      const cp zk = unpack4(xt[lane + 64 * i + 4 * (i >> 2)]);
      const cp zr = unpack4(xt_rev[(64 * 7 + 4) - (64 * i + 4 * (i >> 2))]);
      const cp e = cp{zk.re + zr.re, zk.im - zr.im};
      const cp o = cp{zk.im + zr.im, zr.re - zk.re};
      const float a = 6.283185307179586477f * (float)(64 * i) / (float)NFFT;
      const c32 wk = cmulw(wlo, c32{__builtin_cosf(a), -__builtin_sinf(a)});
      const cp tt = cmulw(o, wk);
      const cp xp = e + tt, xm = e - tt;
      const p2 sp = xp.re * xp.re + xp.im * xp.im, sm = xm.re * xm.re + xm.im * xm.im;
      mg_lo[i] = p2{__builtin_amdgcn_sqrtf(sp.x), __builtin_amdgcn_sqrtf(sp.y)} * bc(0.5f);
      mg_hi[i] = p2{__builtin_amdgcn_sqrtf(sm.x), __builtin_amdgcn_sqrtf(sm.y)} * bc(0.5f);
    }
    const cp z512 = unpack4(xt[512 + 4 * 2]);
    const p2 s512 = z512.re * z512.re + z512.im * z512.im;
    const p2 mg512 = p2{__builtin_amdgcn_sqrtf(s512.x), __builtin_amdgcn_sqrtf(s512.y)};
    SM_FENCE();
#if S2_TIMING
#pragma unroll
    for (int i = 0; i < 8; ++i) asm volatile("" :: "v"(mg_lo[i]), "v"(mg_hi[i]));
    __builtin_amdgcn_sched_barrier(0);
#endif
    S2_STAMP(5);
    // ---- magnitudes over the tile, (t, t+1) pairs per bin (1025 + 3 zero pad)
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      mg2[lane + 64 * i] = mg_lo[i];
      (mg2 + (NC - 64 * 7) - lane)[64 * (7 - i)] = mg_hi[i];
    }
    if (lane == 0) mg2[512] = mg512;
    if (lane >= 1 && lane <= 3) mg2[NC + lane] = bc(0.f);
    SM_FENCE();
    S2_STAMP(6);
#if !S2_EARLY_PREFETCH
    if (fast) load_fast(tn, tnb);                    // next pair's samples travel while the filterbank runs
#endif
    // ---- sparse filterbank: two bands per lane, four bins of both frames per step
    p2 acc0 = bc(0.f), acc1 = bc(0.f);
    const f4* m0 = reinterpret_cast<const f4*>(mg2 + mi.x);
    const f4* m1 = reinterpret_cast<const f4*>(mg2 + mi.z);
    const float4* wq = melw + lane;
#pragma unroll S2_FB_UNROLL
    for (int it = 0; it < P.nit0; ++it) {
      const f4 ma = m0[2 * it], mb = m0[2 * it + 1];
      const float4 ww = wq[it * 64];
      // same order per frame as stft_mel_kernel: w.x mg.x + (w.y mg.y + (w.z mg.z + (w.w mg.w + acc)))
      acc0 = bc(ww.x) * p2{ma.x, ma.y} + (bc(ww.y) * p2{ma.z, ma.w} + (bc(ww.z) * p2{mb.x, mb.y} + (bc(ww.w) * p2{mb.z, mb.w} + acc0)));
    }
    wq += P.nit0 * 64;
#pragma unroll S2_FB_UNROLL
    for (int it = 0; it < P.nit1; ++it) {
      const f4 ma = m1[2 * it], mb = m1[2 * it + 1];
      const float4 ww = wq[it * 64];
      acc1 = bc(ww.x) * p2{ma.x, ma.y} + (bc(ww.y) * p2{ma.z, ma.w} + (bc(ww.z) * p2{mb.x, mb.y} + (bc(ww.w) * p2{mb.z, mb.w} + acc1)));
    }
    SM_FENCE();
#if S2_TIMING
    asm volatile("" :: "v"(acc0), "v"(acc1));
    __builtin_amdgcn_sched_barrier(0);
#endif
    S2_STAMP(7);
    float* out = P.mel_out + ((size_t)b * P.T + t) * P.n_mels;
    if (has0) {
      out[band0] = acc0.x; run_max = fmaxf(run_max, acc0.x);
      if (two) { out[P.n_mels + band0] = acc0.y; run_max = fmaxf(run_max, acc0.y); } else acc0.y = 0.f;
      sq0 = acc0 * acc0 + sq0;
    }
    if (has1) {
      out[band1] = acc1.x; run_max = fmaxf(run_max, acc1.x);
      if (two) { out[P.n_mels + band1] = acc1.y; run_max = fmaxf(run_max, acc1.y); } else acc1.y = 0.f;
      sq1 = acc1 * acc1 + sq1;
    }
    if (more && !fast) load_edge(tn, tnb);
    S2_STAMP(8);
  }
#if S2_TIMING
  if (lane == 0 && blockIdx.y == 77 && (blockIdx.x == 2 || blockIdx.x == 5) && (wv == 1 || wv == 6))
    printf("wave (%d,%d,%d): total %u | load+window %u dft1 %u exch %u dft2 %u dpp %u zwrite+unpack %u mags %u filterbank %u store %u\n",
           blockIdx.x, blockIdx.y, wv, (unsigned)(last_ - first_), ph[0], ph[1], ph[2], ph[3], ph[4], ph[5], ph[6], ph[7], ph[8]);
#endif
  const float wm = wave_max(run_max);
  if (lane == 0 && t_begin < P.T) atomicMax(reinterpret_cast<int*>(P.clip_max + b), __float_as_int(wm));
  SM_FENCE();
  tile[lane] = sq0.x + sq0.y; tile[64 + lane] = sq1.x + sq1.y;
  __syncthreads();
  if (wv == 0) {
    float s0 = 0.f, s1 = 0.f;
#pragma unroll
    for (int u = 0; u < S2_WAVES; ++u) {
      const float* tu = reinterpret_cast<const float*>(tiles + u * S2_TILE_ELEMS);
      s0 += tu[lane]; s1 += tu[64 + lane];
    }
    float* part = P.sumsq_part + ((size_t)b * gridDim.x + blockIdx.x) * P.n_mels;
    if (has0) part[band0] = s0;
    if (has1) part[band1] = s1;
  }
}

// bin_sumsq[b][band] = sum over the workgroup partials of a clip, fixed order
__global__ void mel_sumsq_finish_kernel(const float* __restrict__ part, float* __restrict__ bin_sumsq, int B, int nchunk,
                                        int n_mels) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= B * n_mels) return;
  const int b = i / n_mels, band = i % n_mels;
  float s = 0.f;
  for (int c = 0; c < nchunk; ++c) s += part[((size_t)b * nchunk + c) * n_mels + band];
  bin_sumsq[i] = s;
}

// noisy = x + z * sqrt(mean_t(x^2) * 10^(-snr/10)); also |noisy| clip max for the dB clamp
__global__ void mel_noise_kernel(const float* __restrict__ x, const float* __restrict__ bin_sumsq,
                                 const float* __restrict__ unit_noise, float* __restrict__ out,
                                 float* __restrict__ clip_max, int T, int n_mels, float snr_scale,
                                 uint64_t seed) {
  const int b = blockIdx.y;
  const size_t per = (size_t)T * n_mels;
  float vmax = 0.f;
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < per; i += (size_t)gridDim.x * blockDim.x) {
    const int m = (int)(i % n_mels);
    const float sd = sqrtf(bin_sumsq[(size_t)b * n_mels + m] / (float)T * snr_scale);
    float z;
    const size_t gi = (size_t)b * per + i;
    if (unit_noise) {
      z = unit_noise[gi];
    } else {
      uint4 r = philox4x32(gi >> 1, 0x4e4f4953u, seed);
      const float u1 = ((float)(r.x >> 8) + 0.5f) * (1.0f / 16777216.0f);
      const float u2 = ((float)(r.y >> 8) + 0.5f) * (1.0f / 16777216.0f);
      const float rad = sqrtf(-2.0f * logf(u1));
      float sn, cs;
      sincosf(6.28318530717958647692f * u2, &sn, &cs);
      z = (gi & 1) ? rad * sn : rad * cs;
    }
    const float v = fmaf(z, sd, x[gi]);
    out[gi] = v;
    vmax = fmaxf(vmax, fabsf(v));
  }
  vmax = wave_max(vmax);
  if ((threadIdx.x & 63) == 0) atomicMax(reinterpret_cast<int*>(clip_max + b), __float_as_int(vmax));
}

// per-clip max and per-(clip, band) sum over time of x^2 for features that arrive as LINEAR mel (.npy files of the
// reference) instead of waveforms: one workgroup per clip, thread = band
__global__ __launch_bounds__(256) void mel_stats_kernel(const float* __restrict__ x, float* __restrict__ clip_max,
                                                        float* __restrict__ bin_sumsq, int T, int n_mels) {
  __shared__ float red[256];
  const int b = blockIdx.x, tid = threadIdx.x;
  float mx = 0.f, sq = 0.f;
  if (tid < n_mels)
    for (int t = 0; t < T; ++t) {
      const float v = x[((size_t)b * T + t) * n_mels + tid];
      mx = fmaxf(mx, v);
      sq = fmaf(v, v, sq);
    }
  if (tid < n_mels) bin_sumsq[(size_t)b * n_mels + tid] = sq;
  red[tid] = mx;
  __syncthreads();
  for (int o = 128; o > 0; o >>= 1) {
    if (tid < o) red[tid] = fmaxf(red[tid], red[tid + o]);
    __syncthreads();
  }
  if (tid == 0) clip_max[b] = red[0];
}

// dB + top_db clamp + pad/trunc of the time axis (pad value 0 dB, Transforms.py:89-109)
__global__ void mel_db_kernel(const float* __restrict__ x, const float* __restrict__ clip_max,
                              float* __restrict__ out, int T, int T_out, int n_mels, float top_db) {
  const int b = blockIdx.y;
  const float mx = clip_max[b];
  const float floor_db = 10.0f * log10f(fmaxf(1e-10f, mx * mx)) - top_db;
  const size_t per_out = (size_t)T_out * n_mels;
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < per_out; i += (size_t)gridDim.x * blockDim.x) {
    const int t = (int)(i / n_mels);
    float v = 0.f;
    if (t < T) {
      const float a = x[(size_t)b * T * n_mels + i];
      v = fmaxf(10.0f * log10f(fmaxf(1e-10f, a * a)), floor_db);
    }
    out[(size_t)b * per_out + i] = v;
  }
}

// ---------------------------------------------------------------------------------------------
// C ABI
// ---------------------------------------------------------------------------------------------
extern "C" int bsed_mel_plan_create(const BsedMelCfg* cfg, void** plan_out) {
  BSED_CHECK_ARG(cfg && plan_out, "bsed_mel_plan_create: null argument");
  BSED_CHECK_ARG(cfg->n_fft == NFFT, "bsed_mel_plan_create: only n_fft=2048 is built (got %d)", cfg->n_fft);
  BSED_CHECK_ARG(cfg->n_mels > 0 && cfg->n_mels <= 128, "bsed_mel_plan_create: n_mels must be in 1..128 (two bands per lane)");
  BSED_CHECK_ARG(cfg->hop > 0 && cfg->hop <= NFFT, "bsed_mel_plan_create: bad hop %d", cfg->hop);
  BSED_CHECK_ARG(cfg->fmax <= cfg->sr / 2.0 + 1e-6 && cfg->fmin >= 0 && cfg->fmin < cfg->fmax,
                 "bsed_mel_plan_create: need 0 <= fmin < fmax <= sr/2");
  MelPlan* p = new MelPlan();
  memset(p, 0, sizeof(*p));
  p->cfg = *cfg;
  p->n_bins = NFFT / 2 + 1;
  std::vector<float> win(NFFT);
  // np.hamming: symmetric by construction (numpy evaluates cos(pi n / (M - 1)) on n = 1 - M, 3 - M, ..., M - 1)
  for (int n = 0; n < NFFT; ++n) win[n] = (float)(0.54 - 0.46 * cos(2.0 * M_PI * std::min(n, NFFT - 1 - n) / (NFFT - 1)));
  std::vector<float2> w1(NC), w2(NC + 1);
  for (int k = 0; k < NC; ++k) w1[k] = make_float2((float)cos(-2.0 * M_PI * k / NC), (float)sin(-2.0 * M_PI * k / NC));
  for (int k = 0; k <= NC; ++k) w2[k] = make_float2((float)cos(-2.0 * M_PI * k / NFFT), (float)sin(-2.0 * M_PI * k / NFFT));
  std::vector<int> start, count, off;
  std::vector<float> w;
  build_filterbank(*cfg, start, count, off, w);
  p->nnz = (int)w.size();
  if (w.empty()) w.push_back(0.f);
  BSED_HIP(hipMalloc(&p->d_window, NFFT * sizeof(float)));
  BSED_HIP(hipMalloc(&p->d_w1024, NC * sizeof(float2)));
  BSED_HIP(hipMalloc(&p->d_w2048, (NC + 1) * sizeof(float2)));
  BSED_HIP(hipMalloc(&p->d_mel_start, cfg->n_mels * sizeof(int)));
  BSED_HIP(hipMalloc(&p->d_mel_count, cfg->n_mels * sizeof(int)));
  BSED_HIP(hipMalloc(&p->d_mel_off, cfg->n_mels * sizeof(int)));
  BSED_HIP(hipMalloc(&p->d_mel_w, w.size() * sizeof(float)));
  BSED_HIP(hipMemcpy(p->d_window, win.data(), NFFT * sizeof(float), hipMemcpyHostToDevice));
  BSED_HIP(hipMemcpy(p->d_w1024, w1.data(), NC * sizeof(float2), hipMemcpyHostToDevice));
  BSED_HIP(hipMemcpy(p->d_w2048, w2.data(), (NC + 1) * sizeof(float2), hipMemcpyHostToDevice));
  BSED_HIP(hipMemcpy(p->d_mel_start, start.data(), cfg->n_mels * sizeof(int), hipMemcpyHostToDevice));
  BSED_HIP(hipMemcpy(p->d_mel_count, count.data(), cfg->n_mels * sizeof(int), hipMemcpyHostToDevice));
  BSED_HIP(hipMemcpy(p->d_mel_off, off.data(), cfg->n_mels * sizeof(int), hipMemcpyHostToDevice));
  BSED_HIP(hipMemcpy(p->d_mel_w, w.data(), w.size() * sizeof(float), hipMemcpyHostToDevice));
  // ---- tables of stft_mel_kernel
  std::vector<float2> tw1(15 * 64), tw2(15 * 64), wl(64);
  for (int k = 1; k < 16; ++k)
    for (int l = 0; l < 64; ++l) {
      const double a1 = -2.0 * M_PI * (double)(l * k) / 1024.0, a2 = -2.0 * M_PI * (double)((l & 3) * k) / 64.0;
      tw1[(k - 1) * 64 + l] = make_float2((float)cos(a1), (float)sin(a1));
      tw2[(k - 1) * 64 + l] = make_float2((float)cos(a2), (float)sin(a2));
    }
  for (int l = 0; l < 64; ++l) wl[l] = make_float2((float)cos(-2.0 * M_PI * l / NFFT), (float)sin(-2.0 * M_PI * l / NFFT));
  // filterbank as float4 steps: lane l owns bands l and n_mels-1-l (a long band with a short one)
  const int nm = cfg->n_mels;
  std::vector<int4> midx(64);
  std::vector<int> n4(nm), s4(nm);
  for (int mband = 0; mband < nm; ++mband) {
    s4[mband] = count[mband] ? (start[mband] & ~3) : 0;
    n4[mband] = count[mband] ? (start[mband] + count[mband] - s4[mband] + 3) / 4 : 0;
  }
  int nit0 = 0, nit1 = 0;
  for (int l = 0; l < 64; ++l) {
    const int b0 = l, b1 = nm - 1 - l;
    const bool h0 = b0 < nm && b0 <= b1, h1 = b1 >= 0 && b1 > b0;
    midx[l] = make_int4(h0 ? s4[b0] : 0, h0 ? n4[b0] : 0, h1 ? s4[b1] : 0, h1 ? n4[b1] : 0);
    nit0 = std::max(nit0, midx[l].y);
    nit1 = std::max(nit1, midx[l].w);
  }
  // a padded step reads 4 floats at most 4 * (steps - 1) + 3 past a band's first bin: keep that inside the tile
  for (int l = 0; l < 64; ++l) {
    BSED_CHECK_ARG(midx[l].x + 4 * nit0 <= SM_TILE_FLOATS && midx[l].z + 4 * nit1 <= SM_TILE_FLOATS,
                   "bsed_mel_plan_create: filterbank steps leave the magnitude tile");
  }
  const int nit = nit0 + nit1;
  p->pair_ok = true;
  for (int l = 0; l < 64; ++l)
    if (midx[l].x + 4 * nit0 > 2 * S2_TILE_ELEMS || midx[l].z + 4 * nit1 > 2 * S2_TILE_ELEMS) p->pair_ok = false;
  BSED_CHECK_ARG(nit <= SM_MAX_NIT, "bsed_mel_plan_create: filterbank needs %d float4 steps per lane (max %d)", nit, SM_MAX_NIT);
  std::vector<float4> w4((size_t)nit * 64, make_float4(0.f, 0.f, 0.f, 0.f));
  for (int l = 0; l < 64; ++l)
    for (int which = 0; which < 2; ++which) {
      const int band = which == 0 ? l : nm - 1 - l;
      const int sb = which == 0 ? midx[l].x : midx[l].z, nb = which == 0 ? midx[l].y : midx[l].w;
      const int it0 = which == 0 ? 0 : nit0;
      for (int i = 0; i < nb; ++i) {
        float e[4];
        for (int c = 0; c < 4; ++c) {
          const int bin = sb + 4 * i + c, rel = bin - start[band];
          e[c] = (rel >= 0 && rel < count[band]) ? w[off[band] + rel] : 0.f;
        }
        w4[(size_t)(it0 + i) * 64 + l] = make_float4(e[0], e[1], e[2], e[3]);
      }
    }
  p->nit = nit; p->nit0 = nit0; p->nit1 = nit1;
  BSED_HIP(hipMalloc(&p->d_tw1, tw1.size() * sizeof(float2)));
  BSED_HIP(hipMalloc(&p->d_tw2, tw2.size() * sizeof(float2)));
  BSED_HIP(hipMalloc(&p->d_wl, wl.size() * sizeof(float2)));
  BSED_HIP(hipMalloc(&p->d_melw4, w4.size() * sizeof(float4)));
  BSED_HIP(hipMalloc(&p->d_melidx, midx.size() * sizeof(int4)));
  BSED_HIP(hipMemcpy(p->d_tw1, tw1.data(), tw1.size() * sizeof(float2), hipMemcpyHostToDevice));
  BSED_HIP(hipMemcpy(p->d_tw2, tw2.data(), tw2.size() * sizeof(float2), hipMemcpyHostToDevice));
  BSED_HIP(hipMemcpy(p->d_wl, wl.data(), wl.size() * sizeof(float2), hipMemcpyHostToDevice));
  BSED_HIP(hipMemcpy(p->d_melw4, w4.data(), w4.size() * sizeof(float4), hipMemcpyHostToDevice));
  BSED_HIP(hipMemcpy(p->d_melidx, midx.data(), midx.size() * sizeof(int4), hipMemcpyHostToDevice));
  *plan_out = p;
  return BSED_OK;
}

extern "C" int bsed_mel_plan_destroy(void* plan) {
  if (!plan) return BSED_OK;
  MelPlan* p = (MelPlan*)plan;
  hipFree(p->d_window); hipFree(p->d_w1024); hipFree(p->d_w2048);
  hipFree(p->d_mel_start); hipFree(p->d_mel_count); hipFree(p->d_mel_off); hipFree(p->d_mel_w);
  hipFree(p->d_tw1); hipFree(p->d_tw2); hipFree(p->d_wl); hipFree(p->d_melw4); hipFree(p->d_melidx);
  delete p;
  return BSED_OK;
}

extern "C" int bsed_mel_plan_nnz(const void* plan) { return plan ? ((const MelPlan*)plan)->nnz : -1; }

// two frames per wave (stft_mel2_kernel) whenever its tiles fit the CU's LDS and its padded filterbank steps stay
// inside a wave's tile; BSED_MEL_PAIR=0 keeps the one-frame-per-wave kernel (A/B runs)
static size_t mel_pair_smem(const MelPlan* p) {
  return (size_t)p->nit * 64 * sizeof(float4) + (NC / 2 + 64) * sizeof(float2) + (size_t)S2_WAVES * S2_TILE_ELEMS * sizeof(f4);
}
static bool mel_pair(const MelPlan* p) {
  const char* e = getenv("BSED_MEL_PAIR");   // read per call: tests switch between the two kernels inside one process
  return (e ? atoi(e) : 1) && p->pair_ok && mel_pair_smem(p) <= 160 * 1024;
}
extern "C" int bsed_mel_plan_frames_per_wave(const void* plan) { return plan ? (mel_pair((const MelPlan*)plan) ? 2 : 1) : -1; }

extern "C" int bsed_mel_num_frames(const void* plan, int n_samples) {
  if (!plan || n_samples <= 0) return -1;
  return 1 + n_samples / ((const MelPlan*)plan)->cfg.hop;
}

extern "C" long bsed_mel_scratch_floats(const void* plan, int B, int n_samples) {
  if (!plan || B <= 0 || n_samples <= 0) return -1;
  const MelPlan* p = (const MelPlan*)plan;
  const int T = 1 + n_samples / p->cfg.hop;
  return (long)B * ceil_div(T, SM_WAVES * SM_FPW) * p->cfg.n_mels;
}

extern "C" int bsed_mel_linear(const void* plan, const float* wav, int B, int n_samples, float* mel_lin,
                               float* clip_max, float* bin_sumsq, float* scratch, void* stream) {
  BSED_CHECK_ARG(plan && wav && mel_lin && clip_max && bin_sumsq && scratch, "bsed_mel_linear: null argument");
  const MelPlan* p = (const MelPlan*)plan;
  BSED_CHECK_ARG(B > 0 && B <= 65535, "bsed_mel_linear: B must be in 1..65535");
  BSED_CHECK_ARG(n_samples > NFFT / 2, "bsed_mel_linear: need more than %d samples for reflect padding", NFFT / 2);
  hipStream_t s = (hipStream_t)stream;
  const int T = 1 + n_samples / p->cfg.hop;
  int nchunk = ceil_div(T, SM_WAVES * SM_FPW);
  const size_t smem = (size_t)p->nit * 64 * sizeof(float4) + NC * sizeof(float2) + (size_t)SM_WAVES * SM_TILE_FLOATS * sizeof(float);
  BSED_HIP(hipMemsetAsync(clip_max, 0, (size_t)B * sizeof(float), s));
  SmParams P;
  P.wav = wav; P.n_samples = n_samples; P.hop = p->cfg.hop; P.T = T; P.n_mels = p->cfg.n_mels;
  P.window = p->d_window; P.tw1 = p->d_tw1; P.tw2 = p->d_tw2; P.wl = p->d_wl;
  P.melw4 = p->d_melw4; P.melidx = p->d_melidx; P.nit = p->nit; P.nit0 = p->nit0; P.nit1 = p->nit1;
  P.mel_out = mel_lin; P.clip_max = clip_max; P.sumsq_part = scratch;
  const size_t smem2 = mel_pair_smem(p);
  if (mel_pair(p)) {
    nchunk = ceil_div(T, S2_WAVES * S2_FPW);
    static BsedLdsOnce once2;
    BSED_HIP(bsed_max_lds(once2, (const void*)stft_mel2_kernel));
    hipLaunchKernelGGL(stft_mel2_kernel, dim3(nchunk, B), dim3(S2_THREADS), smem2, s, P);
  } else {
    static BsedLdsOnce once;
    BSED_HIP(bsed_max_lds(once, (const void*)stft_mel_kernel));
    hipLaunchKernelGGL(stft_mel_kernel, dim3(nchunk, B), dim3(MEL_THREADS), smem, s, P);
  }
  hipLaunchKernelGGL(mel_sumsq_finish_kernel, dim3(ceil_div((long)B * P.n_mels, 256)), dim3(256), 0, s, scratch, bin_sumsq, B,
                     nchunk, P.n_mels);
  BSED_LAUNCH_CHECK();
  return BSED_OK;
}

extern "C" int bsed_mel_stats(const float* mel_lin, int B, int T, int n_mels, float* clip_max, float* bin_sumsq,
                              void* stream) {
  BSED_CHECK_ARG(mel_lin && clip_max && bin_sumsq && B > 0 && T > 0 && n_mels > 0 && n_mels <= 256, "bsed_mel_stats: bad argument");
  hipLaunchKernelGGL(mel_stats_kernel, dim3(B), dim3(256), 0, (hipStream_t)stream, mel_lin, clip_max, bin_sumsq, T, n_mels);
  BSED_LAUNCH_CHECK();
  return BSED_OK;
}

extern "C" int bsed_mel_noise(const float* mel_lin, const float* bin_sumsq, const float* unit_noise, int B,
                              int T, int n_mels, float snr_db, uint64_t seed, float* noisy,
                              float* clip_max_noisy, void* stream) {
  BSED_CHECK_ARG(mel_lin && bin_sumsq && noisy && clip_max_noisy, "bsed_mel_noise: null argument");
  BSED_CHECK_ARG(B > 0 && B <= 65535 && T > 0 && n_mels > 0, "bsed_mel_noise: bad shape");
  hipStream_t s = (hipStream_t)stream;
  BSED_HIP(hipMemsetAsync(clip_max_noisy, 0, (size_t)B * sizeof(float), s));
  const size_t per = (size_t)T * n_mels;
  dim3 grid((unsigned)std::min<size_t>(ceil_div(per, 256), 64), B);
  hipLaunchKernelGGL(mel_noise_kernel, grid, dim3(256), 0, s, mel_lin, bin_sumsq, unit_noise, noisy,
                     clip_max_noisy, T, n_mels, powf(10.0f, -snr_db / 10.0f), seed);
  BSED_LAUNCH_CHECK();
  return BSED_OK;
}

extern "C" int bsed_mel_db(const float* mel_lin, const float* clip_max, int B, int T, int T_out, int n_mels,
                           float top_db, float* out_db, void* stream) {
  BSED_CHECK_ARG(mel_lin && clip_max && out_db, "bsed_mel_db: null argument");
  BSED_CHECK_ARG(B > 0 && B <= 65535 && T > 0 && T_out > 0 && n_mels > 0, "bsed_mel_db: bad shape");
  const size_t per = (size_t)T_out * n_mels;
  dim3 grid((unsigned)std::min<size_t>(ceil_div(per, 256), 64), B);
  hipLaunchKernelGGL(mel_db_kernel, grid, dim3(256), 0, (hipStream_t)stream, mel_lin, clip_max, out_db, T,
                     T_out, n_mels, top_db);
  BSED_LAUNCH_CHECK();
  return BSED_OK;
}
