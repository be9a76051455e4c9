// Predictor head + training losses, one workgroup per clip.
//
//   head_fwd_kernel : strong = sigmoid(x Wd^T + bd); p = softmax_class(x Ws^T + bs);
//                     a = clamp(p, 1e-7, 1); weak = sum_t strong*a / sum_t a
//                     [reference src/models/CRNN_GRL.py:441-460, Predictor.forward]
//   head_bwd_kernel : BCE(strong, y) + BCE(weak, y_weak) + w*(MSE(strong, strong_ema) + MSE(weak, weak_ema))
//                     and their gradients back to x, Wd, bd, Ws, bs in one pass
//                     [reference src/main_baseline.py:431-498: nn.BCELoss (log clamped at -100, gradient
//                      (s-y)/max(s(1-s),1e-12)), nn.MSELoss, all reduction='mean']
#include <cstdlib>
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

// row of register r in the 32 x 32 MFMA result fragment (column = lane & 31, lh = lane >> 5)
__device__ __forceinline__ int hd_crow(int r, int lh) { return (r & 3) + 8 * (r >> 2) + 4 * lh; }

// logits for a chunk of frames: lg[f][0..C) dense head, lg[f][C..2C) softmax head.
// The (32 frames x 2C) x 256 contraction runs on v_mfma_f32_32x32x2_f32 (f32 operands, exact fp32 products and sums): with
// scalar FMAs every multiply-add paid an LDS read of the weight (1.2 reads per FMA: the kernel was LDS-bound).  Wave w
// takes column tile w & 1 (classes 32 (w & 1) ..; columns beyond 2C compute on a clamped weight row and are dropped) and
// the K half w >> 1; the two K halves meet in LDS.  Operands: one dword per lane and MFMA, conflict-free (pitch 257).
template <int C>
__device__ __forceinline__ void head_logits(const float* xs /*[HD_FR][HD_K+1]*/, const float* ws /*[2C][HD_K+1]*/,
                                            const float* bs /*[2C]*/, float* lg /*[HD_FR][2C]*/,
                                            float* lgp /*[2][HD_FR][2C] scratch*/, int tid) {
  const int wave = tid >> 6, lane = tid & 63, li = lane & 31, lh = lane >> 5;
  const int nt = wave & 1, kh = wave >> 1;
  const int n = 32 * nt + li;
  const float* ap = xs + li * (HD_K + 1) + (HD_K / 2) * kh + lh;
  const float* bp = ws + min(n, 2 * C - 1) * (HD_K + 1) + (HD_K / 2) * kh + lh;
  f32x16 acc;
#pragma unroll
  for (int r = 0; r < 16; ++r) acc[r] = 0.f;
#pragma unroll 8
  for (int s = 0; s < HD_K / 4; ++s) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(ap[2 * s], bp[2 * s], acc, 0, 0, 0);
  if (n < 2 * C) {
#pragma unroll
    for (int r = 0; r < 16; ++r) lgp[(kh * HD_FR + hd_crow(r, lh)) * (2 * C) + n] = acc[r];
  }
  __syncthreads();
  for (int e = tid; e < HD_FR * 2 * C; e += HD_THREADS) lg[e] = (lgp[e] + lgp[HD_FR * 2 * C + e]) + bs[e % (2 * C)];
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
  float* lgp = sS;                           // [2][32][2C] K-half partials of head_logits: over sS | sA and 32*2C more
  float* exS = sA + HD_FR * C;               // [32][C] exp(logit - max): the second half of lgp (dead after head_logits)
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
    head_logits<C>(xs, ws, bsm, lg, lgp, tid);   // (sS / sA of the previous chunk were consumed before the barrier above)
    __syncthreads();
    // sigmoid / softmax over the classes, one (frame, class) element per thread and round (one THREAD per frame walking
    // its 20 classes -- 40 full-precision exponentials and 40 stores in a dependent chain on 32 of the 256 lanes -- was
    // most of the kernel's time).  Same operations in the same order per element: max, exp, sum over c = 0..C-1, 1/sum.
    float sv[(HD_FR * C + HD_THREADS - 1) / HD_THREADS];
#pragma unroll
    for (int u = 0; u < (HD_FR * C + HD_THREADS - 1) / HD_THREADS; ++u) {
      const int e = tid + u * HD_THREADS;
      if (e < HD_FR * C) {
        const int f = e / C, c = e - f * C;
        float mx = -3.0e38f;
#pragma unroll
        for (int cc = 0; cc < C; ++cc) mx = fmaxf(mx, lg[f * 2 * C + C + cc]);
        exS[e] = expf(lg[f * 2 * C + C + c] - mx);
        sv[u] = sigmoidf_(lg[f * 2 * C + c]);
      }
    }
    __syncthreads();   // (also: every read of lgp's first half -- aliased by sS below -- is long done)
#pragma unroll
    for (int u = 0; u < (HD_FR * C + HD_THREADS - 1) / HD_THREADS; ++u) {
      const int e = tid + u * HD_THREADS;
      if (e < HD_FR * C) {
        const int f = e / C;
        const bool ok = f0 + f < T;
        float se = 0.f;
#pragma unroll
        for (int cc = 0; cc < C; ++cc) se += exS[f * C + cc];
        const float inv = 1.0f / se;
        const float p = exS[e] * inv;
        const float a = attention ? fminf(fmaxf(p, 1e-7f), 1.0f) : 1.0f;
        sS[e] = ok ? sv[u] : 0.f;
        sA[e] = ok ? a : 0.f;
        if (ok) {
          strong[((size_t)b_ * T + f0) * C + e] = sv[u];
          sof_raw[((size_t)b_ * T + f0) * C + e] = p;
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
  float* gaS = lred + HD_THREADS;            // [32][C] attention-path gradient of the softmax output
  float* paS = gaS + HD_FR * C;              // [32][C] softmax output
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
  // The three contractions of the chunk loop run on v_mfma_f32_32x32x2_f32 (f32 operands: exact fp32, like the scalar
  // FMAs they replace, which paid one or two LDS reads per multiply-add).  Wave w owns columns 64 w .. 64 w + 63 of K:
  //   dx[f][k]  = sum_c dl[f][c] W[c][k]        32 frames x 64 columns, 2C / 2 steps
  //   dW[c][k] += sum_f dl[f][c] x[f][k]        2 x 32 class rows (rows >= 2C idle) x 64 columns, 16 steps;
  //                                             the four 32 x 32 accumulators live across the chunks
  const int wave = tid >> 6, lane = tid & 63, li = lane & 31, lh = lane >> 5;
  f32x16 dwa[2][2];
#pragma unroll
  for (int mt = 0; mt < 2; ++mt)
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) dwa[mt][j][r] = 0.f;
  float dbacc = 0.f;  // thread c < 2C accumulates its bias gradient
  const bool has_ys = y_strong != nullptr, has_es = ema_strong != nullptr, has_es2 = ema_strong2 != nullptr,
             has_gx = g_strong_ext != nullptr;
  for (int f0 = f_lo; f0 < f_hi; f0 += HD_FR) {
    __syncthreads();
    head_stage_rows<HD_FR>(xs, x + ((size_t)b_ * T + f0) * HD_K, T - f0, tid);
    // loss gradients, one (frame, class) element per thread and round -- consecutive threads read consecutive addresses,
    // frames past T read the last frame and are masked (no branch around the loads; with one THREAD per frame walking
    // its 20 classes every load waited for the one before it)
    for (int e = tid; e < HD_FR * C; e += HD_THREADS) {
      const int f = e / C, c = e - f * C;
      const bool ok = f0 + f < T;
      const size_t o = ((size_t)b_ * T + min(f0 + f, T - 1)) * C + c;
      const float s = strong[o];
      const float p = sof_raw[o];
      const float ys = has_ys ? y_strong[o] : 0.f;
      const float es = has_es ? ema_strong[o] : 0.f;
      const float es2 = has_es2 ? ema_strong2[o] : 0.f;
      const float gx = has_gx ? g_strong_ext[o] : 0.f;
      const float a = attention ? fminf(fmaxf(p, 1e-7f), 1.0f) : 1.0f;
      float gs = gwk[c] * a / dn[c];
      if (has_ys) {
        gs += w_strong * bce_grad(s, ys) * inv_n_strong;
        l_s += ok ? bce_val(s, ys) : 0.f;
      }
      if (has_es) {
        const float d = s - es;
        gs += w_cons_s * 2.f * d * inv_n_strong;
        l_cs += ok ? d * d : 0.f;
      }
      if (has_es2) {
        const float d = s - es2;
        gs += w_cons_s2 * 2.f * d * inv_n_strong;
        l_cs2 += ok ? d * d : 0.f;
      }
      if (has_gx) gs += gx;
      float ga = 0.f;
      if (attention && p >= 1e-7f && p <= 1.0f) ga = gwk[c] * (s - wk[c]) / dn[c];
      dl[f * 2 * C + c] = ok ? gs * s * (1.f - s) : 0.f;
      gaS[e] = ok ? ga : 0.f;
      paS[e] = ok ? p : 0.f;
    }
    __syncthreads();
    // softmax backward: dl[f][C + c] = p (ga - sum_c' ga p)
    for (int e = tid; e < HD_FR * C; e += HD_THREADS) {
      const int f = e / C;
      float dot = 0.f;
#pragma unroll
      for (int cc = 0; cc < C; ++cc) dot = fmaf(gaS[f * C + cc], paS[f * C + cc], dot);
      dl[f * 2 * C + C + (e - f * C)] = paS[e] * (gaS[e] - dot);
    }
    __syncthreads();
    {
      f32x16 dxa[2];
#pragma unroll
      for (int j = 0; j < 2; ++j)
#pragma unroll
        for (int r = 0; r < 16; ++r) dxa[j][r] = 0.f;
#pragma unroll 4
      for (int st = 0; st < C; ++st) {   // class c = 2 st + lh
        const float av = dl[li * 2 * C + 2 * st + lh];
        const float* wr = ws + (2 * st + lh) * (HD_K + 1) + 64 * wave + li;
        dxa[0] = __builtin_amdgcn_mfma_f32_32x32x2f32(av, wr[0], dxa[0], 0, 0, 0);
        dxa[1] = __builtin_amdgcn_mfma_f32_32x32x2f32(av, wr[32], dxa[1], 0, 0, 0);
      }
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int f = hd_crow(r, lh);
        if (f0 + f < T) {
          float* dxr = dx + ((size_t)b_ * T + f0 + f) * HD_K + 64 * wave + li;
          dxr[0] = dxa[0][r];
          dxr[32] = dxa[1][r];
        }
      }
    }
#pragma unroll 4
    for (int st = 0; st < HD_FR / 2; ++st) {   // frame f = 2 st + lh (frames past T hold dl = 0)
      const int f = 2 * st + lh;
      const float a0 = dl[f * 2 * C + li], a1 = dl[f * 2 * C + min(32 + li, 2 * C - 1)];
      const float* xr = xs + f * (HD_K + 1) + 64 * wave + li;
      const float b0 = xr[0], b1 = xr[32];
      dwa[0][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0, b0, dwa[0][0], 0, 0, 0);
      dwa[0][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0, b1, dwa[0][1], 0, 0, 0);
      dwa[1][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1, b0, dwa[1][0], 0, 0, 0);
      dwa[1][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1, b1, dwa[1][1], 0, 0, 0);
    }
    if (tid < 2 * C)
      for (int f = 0; f < HD_FR; ++f) dbacc += dl[f * 2 * C + tid];
  }
#pragma unroll
  for (int mt = 0; mt < 2; ++mt)
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int c = 32 * mt + hd_crow(r, lh);
        if (c < 2 * C) dw_part[(prow * 2 * C + c) * HD_K + 64 * wave + 32 * j + li] = dwa[mt][j][r];
      }
  if (tid < 2 * C) db_part[prow * 2 * C + tid] = dbacc;
  // loss partials of this clip (plain sums; the host applies weights and 1/N)
  // wave sums (fixed butterfly order), then the four waves' values in order: repeatable
  float vals[6] = {l_s, l_w, l_cs, l_cw, l_cs2, 0.f};
  __syncthreads();
#pragma unroll
  for (int i = 0; i < 5; ++i) {
    const float v = wave_sum(vals[i]);
    if ((tid & 63) == 0) lred[(tid >> 6) * 8 + i] = v;
  }
  __syncthreads();
  if (tid < 6) loss_part[prow * 6 + tid] = tid < 5 ? ((lred[tid] + lred[8 + tid]) + lred[16 + tid]) + lred[24 + tid] : 0.f;
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
  static const int forced = getenv("BSED_HEAD_SPLITS") ? atoi(getenv("BSED_HEAD_SPLITS")) : 0;   // A/B knob
  if (forced > 0) return forced <= chunks ? forced : chunks;
  int S = 1;
  // (B = 256: two splits -- 512 workgroups, one per CU at a time: 85-90 KB of LDS each; four were 15 % slower backward)
  while (S < 8 && (long)B * S < 512 && S * 2 <= chunks) S *= 2;
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
  const size_t smem = (size_t)(2 * C * (HD_K + 1) + HD_FR * (HD_K + 1) + HD_FR * 2 * C + 2 * C + 2 * HD_FR * 2 * C) * 4;
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
  const size_t smem = (size_t)(2 * C * (HD_K + 1) + HD_FR * (HD_K + 1) + HD_FR * 2 * C + 3 * C + HD_THREADS + 2 * HD_FR * C) * 4;
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
