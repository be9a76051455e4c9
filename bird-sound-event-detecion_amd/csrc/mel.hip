// Waveform -> STFT magnitude -> mel -> (noise) -> dB, on gfx950.
//
// Replaces, on the GPU, what the reference does offline on the CPU through librosa:
//   preprocess()                /root/reference/src/data/preprocess.py:18-45
//   AugmentGaussianNoise        /root/reference/src/data/Transforms.py:155-196
//   ApplyLog (amplitude_to_db)  /root/reference/src/data/Transforms.py:74-86
//   PadOrTrunc                  /root/reference/src/data/Transforms.py:89-139
//
// Kernel plan (HBM-bound stage: 1.28 MB wav in + 0.64 MB mel out per 10 s clip at 32 kHz):
//   stft_mel_kernel : one workgroup = 16 consecutive frames of one clip.  The 2048+15*hop sample
//                     span is read ONCE from HBM (coalesced) into LDS with the reflect padding
//                     applied; every frame is windowed out of LDS, transformed by a 1024-point
//                     complex radix-4 Stockham FFT in LDS (real-FFT packing trick), turned into
//                     1025 magnitudes that never leave LDS, and contracted with the sparse
//                     (2016 non-zero) Slaney filterbank.  Only (T,128) linear mel goes to HBM.
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
#define NC 1024  // complex points
#define FPB 8    // frames per workgroup (8: with the magnitudes aliased onto the free FFT buffer the workgroup needs
                 // 39 KB of LDS and four of them fit per CU; the kernel is barrier-latency bound, occupancy is what pays)
#define MEL_THREADS 256

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
__device__ __forceinline__ float2 cmul(float2 a, float2 b) {
  return make_float2(a.x * b.x - a.y * b.y, a.x * b.y + a.y * b.x);
}

__global__ __launch_bounds__(MEL_THREADS) void stft_mel_kernel(
    const float* __restrict__ wav, int n_samples, int hop, int T, int n_mels,
    const float* __restrict__ window, const float2* __restrict__ w1024, const float2* __restrict__ w2048,
    const int* __restrict__ mel_start, const int* __restrict__ mel_count, const int* __restrict__ mel_off,
    const float* __restrict__ mel_w, float* __restrict__ mel_out, float* __restrict__ clip_max,
    float* __restrict__ bin_sumsq, int span_len) {
  extern __shared__ __align__(16) unsigned char smem_raw[];
  float2* bufA = reinterpret_cast<float2*>(smem_raw);  // [1024]
  float2* bufB = bufA + NC;                            // [1024]
  const float2* tw = w1024;                            // twiddles straight from global memory (8 KB, L1-resident)
  float* span = reinterpret_cast<float*>(bufB + NC);   // [span_len]

  const int tid = threadIdx.x;
  const int b = blockIdx.y;
  const int t0 = blockIdx.x * FPB;
  const float* w = wav + (size_t)b * n_samples;

  // stage twiddles + the sample span (reflect padding of librosa.stft(center=True))
  const long g0 = (long)t0 * hop - NFFT / 2;
  for (int i = tid; i < span_len; i += MEL_THREADS) {
    long g = g0 + i;
    if (g < 0) g = -g;
    if (g >= n_samples) g = 2L * (n_samples - 1) - g;
    g = g < 0 ? 0 : (g >= n_samples ? n_samples - 1 : g);
    span[i] = w[g];
  }
  __syncthreads();

  float run_max = 0.f, run_sq = 0.f;
  const int nf = min(FPB, T - t0);
  for (int f = 0; f < nf; ++f) {
    const float* s = span + f * hop;
    // window + pack even/odd samples as one complex sequence
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      const int m = tid + q * MEL_THREADS;
      bufA[m] = make_float2(s[2 * m] * window[2 * m], s[2 * m + 1] * window[2 * m + 1]);
    }
    __syncthreads();
    // 1024-point complex FFT: 5 radix-4 Stockham stages, one butterfly per thread per stage
    float2* src = bufA;
    float2* dst = bufB;
#pragma unroll
    for (int st = 0; st < 5; ++st) {
      const int Ns = 1 << (2 * st);
      const int j = tid;
      const int k = j & (Ns - 1);
      const int tws = (256 / Ns) * k;
      float2 v0 = src[j], v1 = src[j + 256], v2 = src[j + 512], v3 = src[j + 768];
      if (st > 0) {
        v1 = cmul(v1, tw[tws]);
        v2 = cmul(v2, tw[2 * tws]);
        v3 = cmul(v3, tw[3 * tws]);
      }
      const float2 a0 = make_float2(v0.x + v2.x, v0.y + v2.y);
      const float2 a1 = make_float2(v0.x - v2.x, v0.y - v2.y);
      const float2 a2 = make_float2(v1.x + v3.x, v1.y + v3.y);
      const float2 tt = make_float2(v1.x - v3.x, v1.y - v3.y);
      const float2 a3 = make_float2(tt.y, -tt.x);  // * (-i)
      const int j0 = ((j >> (2 * st)) << (2 * st + 2)) + k;
      dst[j0] = make_float2(a0.x + a2.x, a0.y + a2.y);
      dst[j0 + Ns] = make_float2(a1.x + a3.x, a1.y + a3.y);
      dst[j0 + 2 * Ns] = make_float2(a0.x - a2.x, a0.y - a2.y);
      dst[j0 + 3 * Ns] = make_float2(a1.x - a3.x, a1.y - a3.y);
      __syncthreads();
      float2* tmp = src; src = dst; dst = tmp;
    }
    // real-FFT unpack: X[k] = Fe + W2048^k * Fo ; magnitudes into the buffer the last stage left free
    float* mag = reinterpret_cast<float*>(dst);  // 1025 floats of its 2048
    for (int k = tid; k <= NC; k += MEL_THREADS) {
      const float2 zk = src[k & (NC - 1)];
      const float2 zr = src[(NC - k) & (NC - 1)];
      const float2 fe = make_float2(0.5f * (zk.x + zr.x), 0.5f * (zk.y - zr.y));
      const float2 d = make_float2(zk.x - zr.x, zk.y + zr.y);      // Zk - conj(Zr)
      const float2 fo = make_float2(0.5f * d.y, -0.5f * d.x);       // -i/2 * d
      const float2 x = cmul(w2048[k], fo);
      const float re = fe.x + x.x, im = fe.y + x.y;
      mag[k] = sqrtf(re * re + im * im);
    }
    __syncthreads();
    // sparse triangular filterbank: one mel band per thread
    if (tid < n_mels) {
      const int st0 = mel_start[tid], cnt = mel_count[tid];
      const float* mw = mel_w + mel_off[tid];
      float acc = 0.f;
      for (int i = 0; i < cnt; ++i) acc = fmaf(mw[i], mag[st0 + i], acc);
      mel_out[((size_t)b * T + (t0 + f)) * n_mels + tid] = acc;
      run_max = fmaxf(run_max, acc);
      run_sq = fmaf(acc, acc, run_sq);
    }
    // next frame overwrites bufA/mag only after everyone is done with them
    __syncthreads();
  }
  if (tid < n_mels) {
    atomicAdd(&bin_sumsq[(size_t)b * n_mels + tid], run_sq);
    float wm = wave_max(run_max);
    if ((tid & 63) == 0) atomicMax(reinterpret_cast<int*>(clip_max + b), __float_as_int(wm));
  }
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
  BSED_CHECK_ARG(cfg->n_mels > 0 && cfg->n_mels <= MEL_THREADS, "bsed_mel_plan_create: n_mels must be in 1..256");
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
  *plan_out = p;
  return BSED_OK;
}

extern "C" int bsed_mel_plan_destroy(void* plan) {
  if (!plan) return BSED_OK;
  MelPlan* p = (MelPlan*)plan;
  hipFree(p->d_window); hipFree(p->d_w1024); hipFree(p->d_w2048);
  hipFree(p->d_mel_start); hipFree(p->d_mel_count); hipFree(p->d_mel_off); hipFree(p->d_mel_w);
  delete p;
  return BSED_OK;
}

extern "C" int bsed_mel_plan_nnz(const void* plan) { return plan ? ((const MelPlan*)plan)->nnz : -1; }

extern "C" int bsed_mel_num_frames(const void* plan, int n_samples) {
  if (!plan || n_samples <= 0) return -1;
  return 1 + n_samples / ((const MelPlan*)plan)->cfg.hop;
}

extern "C" int bsed_mel_linear(const void* plan, const float* wav, int B, int n_samples, float* mel_lin,
                               float* clip_max, float* bin_sumsq, void* stream) {
  BSED_CHECK_ARG(plan && wav && mel_lin && clip_max && bin_sumsq, "bsed_mel_linear: null argument");
  const MelPlan* p = (const MelPlan*)plan;
  BSED_CHECK_ARG(B > 0 && B <= 65535, "bsed_mel_linear: B must be in 1..65535");
  BSED_CHECK_ARG(n_samples > NFFT / 2, "bsed_mel_linear: need more than %d samples for reflect padding", NFFT / 2);
  hipStream_t s = (hipStream_t)stream;
  const int T = 1 + n_samples / p->cfg.hop;
  const int span_len = NFFT + (FPB - 1) * p->cfg.hop;
  const size_t smem = 2 * NC * sizeof(float2) + (size_t)span_len * sizeof(float);
  BSED_CHECK_ARG(smem <= 160 * 1024, "bsed_mel_linear: hop %d needs %zu B of LDS", p->cfg.hop, smem);
  BSED_HIP(hipMemsetAsync(clip_max, 0, (size_t)B * sizeof(float), s));
  BSED_HIP(hipMemsetAsync(bin_sumsq, 0, (size_t)B * p->cfg.n_mels * sizeof(float), s));
  static BsedLdsOnce once;
  BSED_HIP(bsed_max_lds(once, (const void*)stft_mel_kernel));
  dim3 grid(ceil_div(T, FPB), B);
  hipLaunchKernelGGL(stft_mel_kernel, grid, dim3(MEL_THREADS), smem, s, wav, n_samples, p->cfg.hop, T,
                     p->cfg.n_mels, p->d_window, p->d_w1024, p->d_w2048, p->d_mel_start, p->d_mel_count,
                     p->d_mel_off, p->d_mel_w, mel_lin, clip_max, bin_sumsq, span_len);
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
