// Predictor head + training losses, one workgroup per clip.
//
//   head_fwd_kernel : strong = sigmoid(x Wd^T + bd); p = softmax_class(x Ws^T + bs);
//                     a = clamp(p, 1e-7, 1); weak = sum_t strong*a / sum_t a
//                     [reference src/models/CRNN_GRL.py:441-460, Predictor.forward]
//   head_bwd_kernel : BCE(strong, y) + BCE(weak, y_weak) + w*(MSE(strong, strong_ema) + MSE(weak, weak_ema))
//                     and their gradients back to x, Wd, bd, Ws, bs in one pass
//                     [reference src/main_baseline.py:431-498: nn.BCELoss (log clamped at -100, gradient
//                      (s-y)/max(s(1-s),1e-12)), nn.MSELoss, all reduction='mean']
#include "bsed_common.h"
#include "../../include/bsed.h"

#define HD_K 256       // 2 * n_RNN_cell
#define HD_MAXC 20     // classes per head handled by one thread group
#define HD_FR 32       // frames per chunk
#define HD_THREADS 256

// rows of HD_K floats, global -> LDS rows of HD_K + 1: all of a thread's float4 loads are issued before the first LDS
// store (a load-then-store loop of scalars waits out one memory latency per element: 32 per chunk, ~160 us per clip)
template <int NROWS>
__device__ __forceinline__ void head_stage_rows(float* dst, const float* __restrict__ src, int valid_rows, int tid) {
  constexpr int NV = NROWS * (HD_K / 4) / HD_THREADS;  // float4 per thread
  static_assert(NROWS * (HD_K / 4) % HD_THREADS == 0, "rows must tile the workgroup");
  float4 v[NV];
#pragma unroll
  for (int u = 0; u < NV; ++u) {
    const int e = tid + u * HD_THREADS, r = e / (HD_K / 4), q = e % (HD_K / 4);
    v[u] = r < valid_rows ? *reinterpret_cast<const float4*>(src + (size_t)r * HD_K + 4 * q) : make_float4(0.f, 0.f, 0.f, 0.f);
  }
#pragma unroll
  for (int u = 0; u < NV; ++u) {
    const int e = tid + u * HD_THREADS, r = e / (HD_K / 4), q = e % (HD_K / 4);
    float* d = dst + r * (HD_K + 1) + 4 * q;
    d[0] = v[u].x; d[1] = v[u].y; d[2] = v[u].z; d[3] = v[u].w;
  }
}

// logits for a chunk of frames: lg[f][0..C) dense head, lg[f][C..2C) softmax head
template <int C>
__device__ __forceinline__ void head_logits(const float* xs /*[HD_FR][HD_K+1]*/, const float* ws /*[2C][HD_K+1]*/,
                                            const float* bs /*[2C]*/, float* lg /*[HD_FR][2C]*/, int tid) {
  constexpr int PER = (2 * C) / 8;  // outputs per thread (8 threads per frame)
  const int f = tid >> 3, cg = tid & 7;
  float acc[PER];
#pragma unroll
  for (int i = 0; i < PER; ++i) acc[i] = bs[cg * PER + i];
  for (int k = 0; k < HD_K; ++k) {
    const float xv = xs[f * (HD_K + 1) + k];
#pragma unroll
    for (int i = 0; i < PER; ++i) acc[i] = fmaf(xv, ws[(cg * PER + i) * (HD_K + 1) + k], acc[i]);
  }
#pragma unroll
  for (int i = 0; i < PER; ++i) lg[f * (2 * C) + cg * PER + i] = acc[i];
}

template <int C>
__global__ __launch_bounds__(HD_THREADS) void head_fwd_kernel(
    const float* __restrict__ x, const float* __restrict__ w /*(2C,256): dense rows then softmax rows*/,
    const float* __restrict__ b /*(2C)*/, float* __restrict__ strong, float* __restrict__ sof_raw,
    float* __restrict__ weak, float* __restrict__ den_out, float* __restrict__ part /*(B,S,2,C), S = gridDim.y > 1*/,
    int T, int attention) {
  extern __shared__ __align__(16) float smem[];
  float* ws = smem;                          // [2C][257]
  float* xs = ws + 2 * C * (HD_K + 1);       // [32][257]
  float* lg = xs + HD_FR * (HD_K + 1);       // [32][2C]
  float* bsm = lg + HD_FR * 2 * C;           // [2C]
  float* sS = bsm + 2 * C;                   // [32][C]
  float* sA = sS + HD_FR * C;                // [32][C]
  const int tid = threadIdx.x, b_ = blockIdx.x;
  head_stage_rows<2 * C>(ws, w, 2 * C, tid);
  if (tid < 2 * C) bsm[tid] = b[tid];
  float num = 0.f, den = 0.f;
  // the clip's frames are split over gridDim.y workgroups (whole 32-frame chunks each): one workgroup per clip left
  // the chip at one 4-wave workgroup per CU with ~30 barriers in a row
  const int S = gridDim.y, sp = blockIdx.y;
  const int cps = ((T + HD_FR - 1) / HD_FR + S - 1) / S;
  const int f_lo = sp * cps * HD_FR, f_hi = min(T, (sp + 1) * cps * HD_FR);
  for (int f0 = f_lo; f0 < f_hi; f0 += HD_FR) {
    __syncthreads();
    head_stage_rows<HD_FR>(xs, x + ((size_t)b_ * T + f0) * HD_K, T - f0, tid);
    __syncthreads();
    head_logits<C>(xs, ws, bsm, lg, tid);
    __syncthreads();
    if (tid < HD_FR) {
      const int f = tid;
      const bool ok = f0 + f < T;
      float mx = -3.0e38f;
      for (int c = 0; c < C; ++c) mx = fmaxf(mx, lg[f * 2 * C + C + c]);
      float e[C], se = 0.f;
      for (int c = 0; c < C; ++c) { e[c] = expf(lg[f * 2 * C + C + c] - mx); se += e[c]; }
      const float inv = 1.0f / se;
      for (int c = 0; c < C; ++c) {
        const float s = sigmoidf_(lg[f * 2 * C + c]);
        const float p = e[c] * inv;
        const float a = attention ? fminf(fmaxf(p, 1e-7f), 1.0f) : 1.0f;
        sS[f * C + c] = ok ? s : 0.f;
        sA[f * C + c] = ok ? a : 0.f;
        if (ok) {
          strong[((size_t)b_ * T + f0 + f) * C + c] = s;
          sof_raw[((size_t)b_ * T + f0 + f) * C + c] = p;
        }
      }
    }
    __syncthreads();
    if (tid < C) {
      for (int f = 0; f < HD_FR; ++f) {
        num = fmaf(sS[f * C + tid], sA[f * C + tid], num);
        den += sA[f * C + tid];
      }
    }
  }
  if (tid < C) {
    if (S == 1) {
      weak[(size_t)b_ * C + tid] = num / den;
      den_out[(size_t)b_ * C + tid] = den;
    } else {
      part[(((size_t)b_ * S + sp) * 2 + 0) * C + tid] = num;
      part[(((size_t)b_ * S + sp) * 2 + 1) * C + tid] = den;
    }
  }
}

// weak = sum_s num / sum_s den over the time splits of head_fwd_kernel (fixed order: repeatable)
__global__ void head_weak_kernel(const float* __restrict__ part, float* __restrict__ weak, float* __restrict__ den_out,
                                 int B, int S, int C) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= B * C) return;
  const int b_ = i / C, c = i % C;
  float num = 0.f, den = 0.f;
  for (int s = 0; s < S; ++s) {
    num += part[(((size_t)b_ * S + s) * 2 + 0) * C + c];
    den += part[(((size_t)b_ * S + s) * 2 + 1) * C + c];
  }
  weak[i] = num / den;
  den_out[i] = den;
}

// weak (clip-level) targets of the train loop: out[b][c] = max_t y[b][t][c]  (reference src/main_baseline.py:
// ``target_weak = target.max(-2)[0]`` inside train_mt, once per batch).  One workgroup per clip; a 4.4 MB read that
// torch's generic reduction spends 50 us on (one 64-thread workgroup per five output columns).
#define MT_THREADS 256
__global__ __launch_bounds__(MT_THREADS) void max_over_time_kernel(const float* __restrict__ y, float* __restrict__ out,
                                                                  int T, int C) {
  __shared__ float red[MT_THREADS];
  const int tid = threadIdx.x, b_ = blockIdx.x;
  const int groups = MT_THREADS / C;            // time phases handled side by side
  const int c = tid % C, ph = tid / C;
  float m = -INFINITY;
  if (ph < groups) {
    const float* col = y + (size_t)b_ * T * C + c;
    for (int t = ph; t < T; t += groups) m = fmaxf(m, col[(size_t)t * C]);
  }
  red[tid] = m;
  __syncthreads();
  if (tid < C) {
    for (int g = 1; g < groups; ++g) m = fmaxf(m, red[g * C + tid]);
    out[(size_t)b_ * C + tid] = m;
  }
}

extern "C" int bsed_max_over_time(const float* y, float* out, int B, int T, int C, void* stream) {
  BSED_CHECK_ARG(y && out && B > 0 && T > 0 && C > 0 && C <= MT_THREADS, "bsed_max_over_time: bad argument (C <= %d)", MT_THREADS);
  hipLaunchKernelGGL(max_over_time_kernel, dim3(B), dim3(MT_THREADS), 0, (hipStream_t)stream, y, out, T, C);
  BSED_LAUNCH_CHECK();
  return BSED_OK;
}

// BCE element (PyTorch semantics): value with log clamped at -100
__device__ __forceinline__ float bce_val(float s, float y) {
  return -(y * fmaxf(logf(s), -100.f) + (1.f - y) * fmaxf(logf(1.f - s), -100.f));
}
__device__ __forceinline__ float bce_grad(float s, float y) { return (s - y) / fmaxf((1.f - s) * s, 1e-12f); }

template <int C>
__global__ __launch_bounds__(HD_THREADS) void head_bwd_kernel(
    const float* __restrict__ x, const float* __restrict__ w, const float* __restrict__ strong,
    const float* __restrict__ sof_raw, const float* __restrict__ weak, const float* __restrict__ den,
    const float* __restrict__ y_strong, const float* __restrict__ y_weak, const float* __restrict__ ema_strong,
    const float* __restrict__ ema_weak, const float* __restrict__ ema_strong2, const float* __restrict__ g_strong_ext,
    const float* __restrict__ g_weak_ext, float w_strong, float w_weak, float w_cons_s, float w_cons_w,
    float w_cons_s2, float inv_n_strong, float inv_n_weak,
    float* __restrict__ dx, float* __restrict__ dw_part /*(B,2C,256)*/, float* __restrict__ db_part /*(B,2C)*/,
    float* __restrict__ loss_part /*(B,4)*/, int T, int attention) {
  extern __shared__ __align__(16) float smem[];
  float* ws = smem;                          // [2C][257]
  float* xs = ws + 2 * C * (HD_K + 1);       // [32][257]
  float* dl = xs + HD_FR * (HD_K + 1);       // [32][2C]
  float* gwk = dl + HD_FR * 2 * C;           // [C] d loss / d weak
  float* wk = gwk + C;                       // [C]
  float* dn = wk + C;                        // [C]
  float* lred = dn + C;                      // [HD_THREADS]
  const int tid = threadIdx.x, b_ = blockIdx.x;
  // time splits as in head_fwd_kernel; partial outputs (dW, db, losses) get one row per (clip, split)
  const int S = gridDim.y, sp = blockIdx.y;
  const int cps = ((T + HD_FR - 1) / HD_FR + S - 1) / S;
  const int f_lo = sp * cps * HD_FR, f_hi = min(T, (sp + 1) * cps * HD_FR);
  const size_t prow = (size_t)b_ * S + sp;
  head_stage_rows<2 * C>(ws, w, 2 * C, tid);
  float l_s = 0.f, l_w = 0.f, l_cs = 0.f, l_cw = 0.f, l_cs2 = 0.f;
  if (tid < C) {
    const float wv = weak[(size_t)b_ * C + tid];
    float g = 0.f;
    if (y_weak) {
      const float yv = y_weak[(size_t)b_ * C + tid];
      g += w_weak * bce_grad(wv, yv) * inv_n_weak;
      if (sp == 0) l_w = bce_val(wv, yv);  // clip-level terms are counted by the first split only
    }
    if (ema_weak) {
      const float d = wv - ema_weak[(size_t)b_ * C + tid];
      g += w_cons_w * 2.f * d * inv_n_weak;
      if (sp == 0) l_cw = d * d;
    }
    if (g_weak_ext) g += g_weak_ext[(size_t)b_ * C + tid];
    gwk[tid] = g;
    wk[tid] = wv;
    dn[tid] = den[(size_t)b_ * C + tid];
  }
  float dwacc[2 * C];
#pragma unroll
  for (int c = 0; c < 2 * C; ++c) dwacc[c] = 0.f;
  float dbacc = 0.f;  // thread c < 2C accumulates its bias gradient
  for (int f0 = f_lo; f0 < f_hi; f0 += HD_FR) {
    __syncthreads();
    head_stage_rows<HD_FR>(xs, x + ((size_t)b_ * T + f0) * HD_K, T - f0, tid);
    if (tid < HD_FR) {
      const int f = tid;
      const bool ok = f0 + f < T;
      const size_t o = ((size_t)b_ * T + f0 + f) * C;
      float ga[C], pa[C], dot = 0.f;
      for (int c = 0; c < C; ++c) {
        float dls = 0.f;
        ga[c] = 0.f; pa[c] = 0.f;
        if (ok) {
          const float s = strong[o + c];
          const float p = sof_raw[o + c];
          const float a = attention ? fminf(fmaxf(p, 1e-7f), 1.0f) : 1.0f;
          float gs = gwk[c] * a / dn[c];
          if (y_strong) {
            const float yv = y_strong[o + c];
            gs += w_strong * bce_grad(s, yv) * inv_n_strong;
            l_s += bce_val(s, yv);
          }
          if (ema_strong) {
            const float d = s - ema_strong[o + c];
            gs += w_cons_s * 2.f * d * inv_n_strong;
            l_cs += d * d;
          }
          if (ema_strong2) {
            const float d = s - ema_strong2[o + c];
            gs += w_cons_s2 * 2.f * d * inv_n_strong;
            l_cs2 += d * d;
          }
          if (g_strong_ext) gs += g_strong_ext[o + c];
          dls = gs * s * (1.f - s);
          if (attention && p >= 1e-7f && p <= 1.0f) ga[c] = gwk[c] * (s - wk[c]) / dn[c];
          pa[c] = p;
          dot = fmaf(ga[c], p, dot);
        }
        dl[f * 2 * C + c] = dls;
      }
      for (int c = 0; c < C; ++c) dl[f * 2 * C + C + c] = pa[c] * (ga[c] - dot);
    }
    __syncthreads();
    // dx[f][k]: 8 threads per frame, 32 columns each
    {
      const int f = tid >> 3, kg = tid & 7;
      if (f0 + f < T) {
        float* dxr = dx + ((size_t)b_ * T + f0 + f) * HD_K;
        for (int kk = 0; kk < 32; ++kk) {
          const int k = kg + 8 * kk;
          float a = 0.f;
#pragma unroll
          for (int c = 0; c < 2 * C; ++c) a = fmaf(dl[f * 2 * C + c], ws[c * (HD_K + 1) + k], a);
          dxr[k] = a;
        }
      }
    }
    // dW[c][k = tid] += sum_f dl[f][c] * x[f][k]
    for (int f = 0; f < HD_FR; ++f) {
      const float xv = xs[f * (HD_K + 1) + tid];
#pragma unroll
      for (int c = 0; c < 2 * C; ++c) dwacc[c] = fmaf(dl[f * 2 * C + c], xv, dwacc[c]);
    }
    if (tid < 2 * C)
      for (int f = 0; f < HD_FR; ++f) dbacc += dl[f * 2 * C + tid];
  }
#pragma unroll
  for (int c = 0; c < 2 * C; ++c) dw_part[(prow * 2 * C + c) * HD_K + tid] = dwacc[c];
  if (tid < 2 * C) db_part[prow * 2 * C + tid] = dbacc;
  // loss partials of this clip (plain sums; the host applies weights and 1/N)
  float vals[6] = {l_s, l_w, l_cs, l_cw, l_cs2, 0.f};
  for (int i = 0; i < 6; ++i) {
    __syncthreads();
    lred[tid] = vals[i];
    __syncthreads();
    if (tid == 0) {
      float s = 0.f;
      for (int j = 0; j < HD_THREADS; ++j) s += lred[j];
      loss_part[prow * 6 + i] = s;
    }
  }
}

// Event post-processing of get_predictions (reference src/evaluation_measures.py:188-205): binarise at `threshold`, then
// scipy.ndimage.median_filter(size=(win,1)) along time with scipy's conventions -- window [t - win/2, t - win/2 + win),
// 'reflect' boundary (d c b a | a b c d | d c b a), rank win/2 of the sorted window, i.e. for 0/1 data the output is 1
// iff the window holds at least win - win/2 ones.
__global__ void binarize_median_kernel(const float* __restrict__ strong, float* __restrict__ out, int B, int T, int C,
                                       float threshold, int win) {
  const long total = (long)B * T * C;
  const int need = win - win / 2;
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
    const int c = (int)(i % C);
    const long r = i / C;
    const int t = (int)(r % T);
    const long b = r / T;
    int ones = 0;
    for (int k = 0; k < win; ++k) {
      int tt = t - win / 2 + k;
      // reflect (edge sample repeated): -1 -> 0, -2 -> 1, T -> T-1, T+1 -> T-2; period 2T
      tt %= 2 * T;
      if (tt < 0) tt += 2 * T;
      if (tt >= T) tt = 2 * T - 1 - tt;
      ones += strong[(b * T + tt) * C + c] > threshold ? 1 : 0;
    }
    out[i] = ones >= need ? 1.f : 0.f;
  }
}

extern "C" int bsed_binarize_median(const float* strong, float* out, int B, int T, int C, float threshold, int win,
                                    void* stream) {
  BSED_CHECK_ARG(strong && out && strong != out && B > 0 && T > 0 && C > 0 && win >= 1, "bsed_binarize_median: bad argument");
  const long total = (long)B * T * C;
  hipLaunchKernelGGL(binarize_median_kernel, dim3((unsigned)((total + 255) / 256 > 4096 ? 4096 : (total + 255) / 256)), dim3(256),
                     0, (hipStream_t)stream, strong, out, B, T, C, threshold, win);
  BSED_LAUNCH_CHECK();
  return BSED_OK;
}

// ---------------------------------------------------------------------------------------------
// Contiguous-region decode of the binarised, median-filtered predictions (reference ManyHotEncoder.decode_strong,
// src/utilities/ManyHotEncoder.py:148-164, on dcase_util's find_contiguous_regions: a region starts where the column
// turns on (or at frame 0 if it starts on) and ends where it turns off (or at T)), followed by the frame -> second
// conversion of get_predictions (src/evaluation_measures.py:205-209: frame * pooling_time_ratio / (sr / hop), clipped
// to [0, max_len_seconds], float64 as pandas does it).  One thread per (clip, class) column; two passes (count, then
// write at the exclusive prefix of the counts) so that the event list comes out in the reference's order: clip, then
// class, then time.
// ---------------------------------------------------------------------------------------------
__global__ void decode_count_kernel(const float* __restrict__ mask, int B, int T, int C, int* __restrict__ counts) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= B * C) return;
  const int b = i / C, c = i % C;
  const float* col = mask + (size_t)b * T * C + c;
  int n = 0;
  bool prev = false;
  for (int t = 0; t < T; ++t) {
    const bool on = col[(size_t)t * C] != 0.f;
    n += (on && !prev) ? 1 : 0;
    prev = on;
  }
  counts[i] = n;
}

__global__ void decode_write_kernel(const float* __restrict__ mask, const int* __restrict__ offsets, int B, int T, int C,
                                    double scale, double max_len, int* __restrict__ ev_clip, int* __restrict__ ev_class,
                                    int* __restrict__ ev_frames, double* __restrict__ ev_seconds) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= B * C) return;
  const int b = i / C, c = i % C;
  const float* col = mask + (size_t)b * T * C + c;
  int k = offsets[i];
  bool prev = false;
  int onset = 0;
  for (int t = 0; t <= T; ++t) {
    const bool on = t < T && col[(size_t)t * C] != 0.f;
    if (on && !prev) onset = t;
    if (!on && prev) {
      ev_clip[k] = b; ev_class[k] = c;
      ev_frames[2 * k] = onset; ev_frames[2 * k + 1] = t;
      ev_seconds[2 * k] = fmin(fmax((double)onset * scale, 0.0), max_len);
      ev_seconds[2 * k + 1] = fmin(fmax((double)t * scale, 0.0), max_len);
      ++k;
    }
    prev = on;
  }
}

extern "C" int bsed_decode_count(const float* mask, int B, int T, int C, int* counts, void* stream) {
  BSED_CHECK_ARG(mask && counts && B > 0 && T > 0 && C > 0, "bsed_decode_count: bad argument");
  hipLaunchKernelGGL(decode_count_kernel, dim3((B * C + 127) / 128), dim3(128), 0, (hipStream_t)stream, mask, B, T, C, counts);
  BSED_LAUNCH_CHECK();
  return BSED_OK;
}

extern "C" int bsed_decode_write(const float* mask, const int* offsets, int B, int T, int C, double scale, double max_len,
                                 int* ev_clip, int* ev_class, int* ev_frames, double* ev_seconds, void* stream) {
  BSED_CHECK_ARG(mask && offsets && ev_clip && ev_class && ev_frames && ev_seconds && B > 0 && T > 0 && C > 0,
                 "bsed_decode_write: bad argument");
  hipLaunchKernelGGL(decode_write_kernel, dim3((B * C + 127) / 128), dim3(128), 0, (hipStream_t)stream, mask, offsets, B, T,
                     C, scale, max_len, ev_clip, ev_class, ev_frames, ev_seconds);
  BSED_LAUNCH_CHECK();
  return BSED_OK;
}

// time splits per clip: enough workgroups for ~4 per CU, whole 32-frame chunks each
extern "C" int bsed_head_splits(int B, int T) {
  const int chunks = (T + HD_FR - 1) / HD_FR;
  int S = 1;
  while (S < 8 && (long)B * S < 1024 && S * 2 <= chunks) S *= 2;
  return S;
}

extern "C" int bsed_head_fwd(const float* x, const float* w, const float* b, float* strong, float* sof_raw,
                             float* weak, float* den, float* part, int B, int T, int K, int C, int attention,
                             void* stream) {
  BSED_CHECK_ARG(x && w && b && strong && sof_raw && weak && den, "bsed_head_fwd: null tensor");
  const int S = bsed_head_splits(B, T);
  BSED_CHECK_ARG(S == 1 || part, "bsed_head_fwd: %d time splits need the (B,%d,2,C) scratch buffer", S, S);
  BSED_CHECK_ARG(B > 0 && T > 0, "bsed_head_fwd: bad shape");
  BSED_CHECK_ARG(K == HD_K && C == 20, "bsed_head_fwd: built for K=256, nclass=20 (got %d, %d)", K, C);
  const size_t smem = (size_t)(2 * C * (HD_K + 1) + HD_FR * (HD_K + 1) + HD_FR * 2 * C + 2 * C + 2 * HD_FR * C) * 4;
  static BsedLdsOnce once;
  BSED_HIP(bsed_max_lds(once, (const void*)head_fwd_kernel<20>));
  hipLaunchKernelGGL(head_fwd_kernel<20>, dim3(B, S), dim3(HD_THREADS), smem, (hipStream_t)stream, x, w, b, strong,
                     sof_raw, weak, den, part, T, attention);
  if (S > 1)
    hipLaunchKernelGGL(head_weak_kernel, dim3((B * C + 255) / 256), dim3(256), 0, (hipStream_t)stream, part, weak, den, B,
                       S, C);
  BSED_LAUNCH_CHECK();
  return BSED_OK;
}

extern "C" int bsed_head_bwd(const BsedHeadBwdDesc* d, void* stream) {
  BSED_CHECK_ARG(d, "bsed_head_bwd: null descriptor");
  BSED_CHECK_ARG(d->x && d->w && d->strong && d->sof_raw && d->weak && d->den && d->dx && d->dw_part && d->db_part &&
                     d->loss_part, "bsed_head_bwd: null tensor");
  BSED_CHECK_ARG(d->B > 0 && d->T > 0 && d->K == HD_K && d->C == 20, "bsed_head_bwd: built for K=256, nclass=20");
  const int C = 20;
  const size_t smem = (size_t)(2 * C * (HD_K + 1) + HD_FR * (HD_K + 1) + HD_FR * 2 * C + 3 * C + HD_THREADS) * 4;
  static BsedLdsOnce once;
  BSED_HIP(bsed_max_lds(once, (const void*)head_bwd_kernel<20>));
  hipLaunchKernelGGL(head_bwd_kernel<20>, dim3(d->B, bsed_head_splits(d->B, d->T)), dim3(HD_THREADS), smem,
                     (hipStream_t)stream, d->x, d->w,
                     d->strong, d->sof_raw, d->weak, d->den, d->y_strong, d->y_weak, d->ema_strong, d->ema_weak,
                     d->ema_strong2, d->g_strong_ext, d->g_weak_ext, d->w_strong, d->w_weak, d->w_cons_s, d->w_cons_w,
                     d->w_cons_s2, d->inv_n_strong, d->inv_n_weak, d->dx, d->dw_part, d->db_part, d->loss_part, d->T,
                     d->attention);
  BSED_LAUNCH_CHECK();
  return BSED_OK;
}
