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
  for (int n = 0; n < NFFT; ++n) win[n] = (float)(0.54 - 0.46 * cos(2.0 * M_PI * n / (NFFT - 1)));  // np.hamming
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
  const int nchunk = ceil_div(T, SM_WAVES * SM_FPW);
  const size_t smem = (size_t)p->nit * 64 * sizeof(float4) + NC * sizeof(float2) + (size_t)SM_WAVES * SM_TILE_FLOATS * sizeof(float);
  BSED_HIP(hipMemsetAsync(clip_max, 0, (size_t)B * sizeof(float), s));
  SmParams P;
  P.wav = wav; P.n_samples = n_samples; P.hop = p->cfg.hop; P.T = T; P.n_mels = p->cfg.n_mels;
  P.window = p->d_window; P.tw1 = p->d_tw1; P.tw2 = p->d_tw2; P.wl = p->d_wl;
  P.melw4 = p->d_melw4; P.melidx = p->d_melidx; P.nit = p->nit; P.nit0 = p->nit0; P.nit1 = p->nit1;
  P.mel_out = mel_lin; P.clip_max = clip_max; P.sumsq_part = scratch;
  static BsedLdsOnce once;
  BSED_HIP(bsed_max_lds(once, (const void*)stft_mel_kernel));
  hipLaunchKernelGGL(stft_mel_kernel, dim3(nchunk, B), dim3(MEL_THREADS), smem, s, P);
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
