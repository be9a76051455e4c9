// CNN pieces that are HBM-bound streaming passes rather than contractions.
//
//   conv0_fwd_kernel     first conv (Cin = 1): a 9-tap stencil, no MFMA          [src/models/CNN.py:46-47, i = 0]
//   conv0_wgrad_kernel   its weight gradient (thread-private 9 x CO accumulators)
//   stats_chunk / stats_finish  per-tile partial sums -> fp64 chunk sums -> totals + BatchNorm finalize (scale/shift,
//                        running stats) / BatchNorm-backward coefficients / bias gradients
//                                                                               [src/models/CNN.py:49]
//   bn_eval_kernel       eval-mode scale/shift from the running statistics
//   bn_bwd_apply         BatchNorm backward as one affine map d_y = A g + B (y-mean) + C
//   colsum / bias helpers
#include "bsed_common.h"
#include "../../include/bsed.h"
#include <algorithm>

// ---------------------------------------------------------------------------------------------
template <int CO>
__global__ __launch_bounds__(256) void conv0_fwd_kernel(const float* __restrict__ x, const float* __restrict__ w,
                                                        const float* __restrict__ bias, float* __restrict__ y,
                                                        float* __restrict__ stats, int H, int W, long per_img) {
  __shared__ float ws[CO * 9 + CO];
  __shared__ float sv[256 * (CO + 1)];
  const int tid = threadIdx.x;
  for (int i = tid; i < CO * 9 + CO; i += 256) ws[i] = i < CO * 9 ? w[i] : bias[i - CO * 9];
  __syncthreads();
  const int nb = blockIdx.y;
  const long pos = (long)blockIdx.x * 256 + tid;
  const bool ok = pos < per_img;
  float out[CO];
#pragma unroll
  for (int c = 0; c < CO; ++c) out[c] = 0.f;
  if (ok) {
    const int h = (int)(pos / W), wq = (int)(pos % W);
    const float* xi = x + (size_t)nb * per_img;
    float in9[9];
#pragma unroll
    for (int kh = 0; kh < 3; ++kh)
#pragma unroll
      for (int kw = 0; kw < 3; ++kw) {
        const int hh = h + kh - 1, ww = wq + kw - 1;
        in9[kh * 3 + kw] = (hh >= 0 && hh < H && ww >= 0 && ww < W) ? xi[(size_t)hh * W + ww] : 0.f;
      }
#pragma unroll
    for (int c = 0; c < CO; ++c) {
      float a = ws[CO * 9 + c];
#pragma unroll
      for (int t = 0; t < 9; ++t) a = fmaf(in9[t], ws[c * 9 + t], a);
      out[c] = a;
    }
  }
  // The outputs go through LDS so that a store instruction writes 1 KB of consecutive addresses (lane = consecutive
  // float4 of the workgroup's [256 positions][CO] block) instead of 16 bytes out of every thread's CO*4-byte row.
#pragma unroll
  for (int c = 0; c < CO; ++c) sv[tid * (CO + 1) + c] = out[c];  // zeros for out-of-range positions
  __syncthreads();
  {
    const long pos0 = (long)blockIdx.x * 256;
    float4* dst = reinterpret_cast<float4*>(y + ((size_t)nb * per_img + pos0) * CO);
    constexpr int Q = CO / 4;  // float4 per position
#pragma unroll
    for (int k = 0; k < Q; ++k) {
      const int e = tid + 256 * k, p = e / Q, q = e % Q;
      if (pos0 + p < per_img) {
        const float* r = sv + p * (CO + 1) + 4 * q;
        dst[e] = make_float4(r[0], r[1], r[2], r[3]);
      }
    }
  }
  if (stats) {
    // thread (c, g): channel c over positions g*PG .. g*PG+PG-1
    constexpr int NG = 256 / CO;   // groups
    constexpr int PG = 256 / NG;   // positions per group (= CO)
    {
      const int c = tid % CO, g = tid / CO;
      float s = 0.f, q = 0.f;
      for (int i = 0; i < PG; ++i) {
        const float v = sv[(g * PG + i) * (CO + 1) + c];
        s += v;
        q = fmaf(v, v, q);
      }
      __syncthreads();
      sv[tid] = s;
      sv[256 + tid] = q;
    }
    __syncthreads();
    if (tid < 2 * CO) {
      const int which = tid / CO, c = tid % CO;
      float s = 0.f;
      for (int g = 0; g < NG; ++g) s += sv[which * 256 + g * CO + c];
      const long blk = (long)nb * gridDim.x + blockIdx.x;
      stats[(blk * 2 + which) * CO + c] = s;
    }
  }
}

// dW0[co][tap] partials: part[blk][tap][co].
// BNB: dy is the gradient w.r.t. the BatchNorm OUTPUT and the BatchNorm backward d_y = A g + B (y - mean) + C
// (coef = [A | B | C] from bn_bwd_finalize) is applied on load, so the first block's d_y (the largest tensor of the
// network) is never written to or re-read from HBM.
template <int CO, bool BNB>
__global__ __launch_bounds__(256) void conv0_wgrad_kernel(const float* __restrict__ x, const float* __restrict__ dy,
                                                          const float* __restrict__ y, const float* __restrict__ coef,
                                                          const float* __restrict__ mean, float* __restrict__ part,
                                                          int NB, int H, int W) {
  __shared__ float sv[256 * (CO + 1)];
  const int tid = threadIdx.x;
  const long per_img = (long)H * W, total = per_img * NB;
  float acc[9][CO];
#pragma unroll
  for (int t = 0; t < 9; ++t)
#pragma unroll
    for (int c = 0; c < CO; ++c) acc[t][c] = 0.f;
  for (long p = (long)blockIdx.x * 256 + tid; p < total; p += (long)gridDim.x * 256) {
    const long nb = p / per_img, q = p - nb * per_img;
    const int h = (int)(q / W), wq = (int)(q % W);
    const float* xi = x + nb * per_img;
    float g[CO];
    const float4* src = reinterpret_cast<const float4*>(dy + (size_t)p * CO);
#pragma unroll
    for (int c = 0; c < CO; c += 4) {
      const float4 v = src[c / 4];
      g[c] = v.x; g[c + 1] = v.y; g[c + 2] = v.z; g[c + 3] = v.w;
    }
    if (BNB) {
      const float4* ys = reinterpret_cast<const float4*>(y + (size_t)p * CO);
#pragma unroll
      for (int c = 0; c < CO; c += 4) {
        const float4 v = ys[c / 4];
        const float yv[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
        for (int i = 0; i < 4; ++i)
          g[c + i] = fmaf(coef[c + i], g[c + i], fmaf(coef[CO + c + i], yv[i] - mean[c + i], coef[2 * CO + c + i]));
      }
    }
#pragma unroll
    for (int kh = 0; kh < 3; ++kh)
#pragma unroll
      for (int kw = 0; kw < 3; ++kw) {
        const int hh = h + kh - 1, ww = wq + kw - 1;
        const float xv = (hh >= 0 && hh < H && ww >= 0 && ww < W) ? xi[(size_t)hh * W + ww] : 0.f;
#pragma unroll
        for (int c = 0; c < CO; ++c) acc[kh * 3 + kw][c] = fmaf(xv, g[c], acc[kh * 3 + kw][c]);
      }
  }
  constexpr int NG = 256 / CO, PG = CO;
#pragma unroll
  for (int t = 0; t < 9; ++t) {
    __syncthreads();
#pragma unroll
    for (int c = 0; c < CO; ++c) sv[tid * (CO + 1) + c] = acc[t][c];
    __syncthreads();
    const int c = tid % CO, g = tid / CO;
    float s = 0.f;
    for (int i = 0; i < PG; ++i) s += sv[(g * PG + i) * (CO + 1) + c];
    __syncthreads();
    sv[tid] = s;
    __syncthreads();
    if (tid < CO) {
      float r = 0.f;
      for (int gg = 0; gg < NG; ++gg) r += sv[gg * CO + tid];
      part[((size_t)blockIdx.x * 9 + t) * CO + tid] = r;
    }
  }
}

__global__ void bn_eval_kernel(int C, float eps, const float* __restrict__ gamma, const float* __restrict__ beta,
                               const float* __restrict__ running_mean, const float* __restrict__ running_var,
                               float* __restrict__ scale, float* __restrict__ shift) {
  const int c = blockIdx.x * blockDim.x + threadIdx.x;
  if (c >= C) return;
  const float sc = gamma[c] / sqrtf(running_var[c] + eps);
  scale[c] = sc;
  shift[c] = beta[c] - running_mean[c] * sc;
}

// the eval-mode scale / shift of every BatchNorm layer of a forward in one launch (blockIdx.y = layer)
struct BnEvalBatch {
  struct { const float *gamma, *beta, *rmean, *rvar; float *scale, *shift; int C; } j[BSED_BN_EVAL_MAX_JOBS];
  float eps;
};
__global__ void bn_eval_batch_kernel(const BnEvalBatch Bt) {
  const auto& J = Bt.j[blockIdx.y];
  const int c = blockIdx.x * blockDim.x + threadIdx.x;
  if (c >= J.C) return;
  const float sc = J.gamma[c] / sqrtf(J.rvar[c] + Bt.eps);   // same expression as bn_eval_kernel: same bits
  J.scale[c] = sc;
  J.shift[c] = J.beta[c] - J.rmean[c] * sc;
}

// sums = (sum g, sum g*y) -> dgamma, dbeta and the coefficients of d_y = A g + B (y - mean) + Cc
__global__ void bn_bwd_apply_kernel(float* __restrict__ g, const float* __restrict__ y, const float* __restrict__ coef,
                                    const float* __restrict__ mean, long n4, int C) {
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += (long)gridDim.x * blockDim.x) {
    const int c = (int)((i * 4) % C);
    float4 gv = reinterpret_cast<float4*>(g)[i];
    const float4 yv = reinterpret_cast<const float4*>(y)[i];
    const float4 A = *reinterpret_cast<const float4*>(coef + c);
    const float4 B = *reinterpret_cast<const float4*>(coef + C + c);
    const float4 Cc = *reinterpret_cast<const float4*>(coef + 2 * C + c);
    const float4 m = *reinterpret_cast<const float4*>(mean + c);
    gv.x = fmaf(A.x, gv.x, fmaf(B.x, yv.x - m.x, Cc.x));
    gv.y = fmaf(A.y, gv.y, fmaf(B.y, yv.y - m.y, Cc.y));
    gv.z = fmaf(A.z, gv.z, fmaf(B.z, yv.z - m.z, Cc.z));
    gv.w = fmaf(A.w, gv.w, fmaf(B.w, yv.w - m.w, Cc.w));
    reinterpret_cast<float4*>(g)[i] = gv;
  }
}

// ---------------------------------------------------------------------------------------------
// Reduction + finalize in TWO launches: the chunk workgroups write fp64 partial sums, then one workgroup per 32
// channels adds the chunks in a fixed order and runs the finalize step (BatchNorm scale/shift + running statistics,
// BatchNorm-backward coefficients, or a bias gradient) in the same kernel.  (A single launch whose last workgroup
// finishes was measured 4x SLOWER: the device-scope release fence writes back the whole L2, which is full of the
// activations the producer kernel just wrote.)
enum { FIN_BN_FWD = 0, FIN_BN_BWD = 1, FIN_TO_GRAD = 2 };
struct StatsFin {
  int mode, which, accumulate;
  double count; float eps, momentum;
  const float* gamma; const float* beta; const float* mean; const float* invstd;
  float* running_mean; float* running_var; long long* nbt;
  float* mean_out; float* invstd_out; float* scale; float* shift;
  float* dgamma; float* dbeta; float* coef; float* dst;
};

__device__ __forceinline__ void stats_finish(const StatsFin& F, int C, int c, double s0, double s1) {
  if (F.mode == FIN_BN_FWD) {
    if (c == 0 && F.nbt) *F.nbt += 1;
    const double mean = s0 / F.count;
    double var = s1 / F.count - mean * mean;
    if (var < 0) var = 0;
    const double invstd = 1.0 / sqrt(var + (double)F.eps);
    F.mean_out[c] = (float)mean;
    F.invstd_out[c] = (float)invstd;
    const float sc = F.gamma[c] * (float)invstd;
    F.scale[c] = sc;
    F.shift[c] = F.beta[c] - (float)mean * sc;
    if (F.running_mean) {
      const double unbiased = F.count > 1 ? var * F.count / (F.count - 1) : var;
      F.running_mean[c] = (1.f - F.momentum) * F.running_mean[c] + F.momentum * (float)mean;
      F.running_var[c] = (1.f - F.momentum) * F.running_var[c] + F.momentum * (float)unbiased;
    }
  } else if (F.mode == FIN_BN_BWD) {
    const double sg = s0, sgy = s1;
    const double m = F.mean[c], is = F.invstd[c];
    const double dgam = (sgy - m * sg) * is;
    const double A = (double)F.gamma[c] * is;
    F.coef[c] = (float)A;
    F.coef[C + c] = (float)(-A * is * dgam / F.count);
    F.coef[2 * C + c] = (float)(-A * sg / F.count);
    if (F.accumulate) { F.dgamma[c] += (float)dgam; F.dbeta[c] += (float)sg; }
    else { F.dgamma[c] = (float)dgam; F.dbeta[c] = (float)sg; }
  } else {
    const float v = (float)(F.which ? s1 : s0);
    F.dst[c] = F.accumulate ? F.dst[c] + v : v;
  }
}

__global__ __launch_bounds__(256) void stats_chunk_kernel(const float* __restrict__ partial, long ntiles, int C,
                                                          double* __restrict__ chunks) {
  __shared__ double sm[2][256];
  const int tid = threadIdx.x;
  const int cl = tid & 31, g = tid >> 5;  // 32 channels x 8 tile groups
  const int c = blockIdx.y * 32 + cl;
  const long chunk = blockIdx.x, nchunks = gridDim.x;
  const long per = (ntiles + nchunks - 1) / nchunks;
  const long t0 = chunk * per, t1 = (t0 + per < ntiles) ? t0 + per : ntiles;
  double s0 = 0.0, s1 = 0.0;
  if (c < C) {
    // four tiles per trip, all eight loads issued before the first add: the loop is a chain of L2 latencies otherwise
    // (fixed summation order: ((t, t+8), (t+16, t+24)) per trip, trips in order)
    long t = t0 + g;
    for (; t + 24 < t1; t += 32) {
      float a[4], b[4];
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        a[u] = partial[((t + 8 * u) * 2 + 0) * C + c];
        b[u] = partial[((t + 8 * u) * 2 + 1) * C + c];
      }
      s0 += ((double)a[0] + (double)a[1]) + ((double)a[2] + (double)a[3]);
      s1 += ((double)b[0] + (double)b[1]) + ((double)b[2] + (double)b[3]);
    }
    for (; t < t1; t += 8) {
      s0 += (double)partial[(t * 2 + 0) * C + c];
      s1 += (double)partial[(t * 2 + 1) * C + c];
    }
  }
  sm[0][tid] = s0; sm[1][tid] = s1;
  __syncthreads();
  if (g == 0 && c < C) {
    double r0 = 0.0, r1 = 0.0;
    for (int gg = 0; gg < 8; ++gg) { r0 += sm[0][gg * 32 + cl]; r1 += sm[1][gg * 32 + cl]; }
    chunks[(chunk * 2 + 0) * C + c] = r0;
    chunks[(chunk * 2 + 1) * C + c] = r1;
  }
}

__global__ __launch_bounds__(256) void stats_finish_kernel(const double* __restrict__ chunks, int nchunks, int C,
                                                           const StatsFin F) {
  __shared__ double sm[2][256];
  const int tid = threadIdx.x;
  const int cl = tid & 31, g = tid >> 5;
  const int c = blockIdx.x * 32 + cl;
  double s0 = 0.0, s1 = 0.0;
  if (c < C) {
    int k = g;
    for (; k + 24 < nchunks; k += 32) {   // four chunks per trip, loads first (see stats_chunk_kernel)
      double a[4], b[4];
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        a[u] = chunks[((size_t)(k + 8 * u) * 2 + 0) * C + c];
        b[u] = chunks[((size_t)(k + 8 * u) * 2 + 1) * C + c];
      }
      s0 += (a[0] + a[1]) + (a[2] + a[3]);
      s1 += (b[0] + b[1]) + (b[2] + b[3]);
    }
    for (; k < nchunks; k += 8) {
      s0 += chunks[((size_t)k * 2 + 0) * C + c];
      s1 += chunks[((size_t)k * 2 + 1) * C + c];
    }
  }
  sm[0][tid] = s0; sm[1][tid] = s1;
  __syncthreads();
  if (g == 0 && c < C) {
    double r0 = 0.0, r1 = 0.0;
    for (int gg = 0; gg < 8; ++gg) { r0 += sm[0][gg * 32 + cl]; r1 += sm[1][gg * 32 + cl]; }
    stats_finish(F, C, c, r0, r1);
  }
}

// per-block column sums of a (M, C) matrix with row pitch: part[blk][2][C] (slot 1 = 0), fed to stats_reduce
__global__ __launch_bounds__(256) void colsum_kernel(const float* __restrict__ in, long M, int C, int pitch,
                                                     float* __restrict__ part) {
  __shared__ float sm[256];
  const int tid = threadIdx.x, cl = tid & 31, g = tid >> 5;
  const long per = (M + gridDim.x - 1) / gridDim.x;
  const long r0 = (long)blockIdx.x * per, r1 = (r0 + per < M) ? r0 + per : M;
  for (int cb = 0; cb < C; cb += 32) {
    const int c = cb + cl;
    float s = 0.f;
    if (c < C) {
      long r = r0 + g;
      for (; r + 24 < r1; r += 32) {   // four rows per trip, loads first
        const float a0 = in[r * pitch + c], a1 = in[(r + 8) * pitch + c], a2 = in[(r + 16) * pitch + c],
                    a3 = in[(r + 24) * pitch + c];
        s += (a0 + a1) + (a2 + a3);
      }
      for (; r < r1; r += 8) s += in[r * pitch + c];
    }
    __syncthreads();
    sm[tid] = s;
    __syncthreads();
    if (g == 0 && c < C) {
      float t = 0.f;
      for (int gg = 0; gg < 8; ++gg) t += sm[gg * 32 + cl];
      part[((size_t)blockIdx.x * 2 + 0) * C + c] = t;
      part[((size_t)blockIdx.x * 2 + 1) * C + c] = 0.f;
    }
  }
}

// elementwise dropout: out = in * keep/(1-p)   (forward and backward use the same call)
__global__ void dropout_kernel(const float* __restrict__ in, float* __restrict__ out, long n, float p,
                               uint32_t rng_stream, uint64_t seed, const uint64_t* __restrict__ seed_add) {
  if (seed_add) seed += *seed_add;
  const uint32_t key = drop_key(rng_stream, seed), thr = drop_threshold(p);
  const float sc = p > 0.f ? 1.0f / (1.0f - p) : 1.0f;
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x)
    out[i] = in[i] * drop_mul((uint64_t)i, key, thr, sc);
}

// ---------------------------------------------------------------------------------------------
// C ABI
// ---------------------------------------------------------------------------------------------
extern "C" int bsed_conv0_fwd(const float* x, const float* w, const float* bias, float* y, float* stats, int NB, int H,
                              int W, int CO, void* stream) {
  BSED_CHECK_ARG(x && w && bias && y, "bsed_conv0_fwd: null tensor");
  BSED_CHECK_ARG(NB > 0 && NB <= 65535 && H > 0 && W > 0, "bsed_conv0_fwd: bad shape");
  const long per = (long)H * W;
  dim3 grid(ceil_div(per, 256), NB);
  hipStream_t s = (hipStream_t)stream;
  if (CO == 16) hipLaunchKernelGGL(conv0_fwd_kernel<16>, grid, dim3(256), 0, s, x, w, bias, y, stats, H, W, per);
  else if (CO == 32) hipLaunchKernelGGL(conv0_fwd_kernel<32>, grid, dim3(256), 0, s, x, w, bias, y, stats, H, W, per);
  else { bsed_set_error("bsed_conv0_fwd: first-layer width %d not built (16 or 32)", CO); return BSED_ERR_ARG; }
  BSED_LAUNCH_CHECK();
  return BSED_OK;
}

extern "C" int bsed_conv0_num_tiles(int NB, int H, int W) { return NB * ceil_div((long)H * W, 256); }

extern "C" int bsed_conv0_wgrad(const float* x, const float* dy, const float* y, const float* coef, const float* mean,
                                float* part, int G, int NB, int H, int W, int CO,
                                void* stream) {
  BSED_CHECK_ARG(x && dy && part && G > 0 && NB > 0 && H > 0 && W > 0, "bsed_conv0_wgrad: bad argument");
  hipStream_t s = (hipStream_t)stream;
  BSED_CHECK_ARG((y == nullptr) == (coef == nullptr) && (y == nullptr) == (mean == nullptr),
                 "bsed_conv0_wgrad: y, coef and mean come together (BatchNorm backward applied on load) or not at all");
  if (CO == 16 && y) hipLaunchKernelGGL((conv0_wgrad_kernel<16, true>), dim3(G), dim3(256), 0, s, x, dy, y, coef, mean, part, NB, H, W);
  else if (CO == 16) hipLaunchKernelGGL((conv0_wgrad_kernel<16, false>), dim3(G), dim3(256), 0, s, x, dy, y, coef, mean, part, NB, H, W);
  else if (CO == 32 && y) hipLaunchKernelGGL((conv0_wgrad_kernel<32, true>), dim3(G), dim3(256), 0, s, x, dy, y, coef, mean, part, NB, H, W);
  else if (CO == 32) hipLaunchKernelGGL((conv0_wgrad_kernel<32, false>), dim3(G), dim3(256), 0, s, x, dy, y, coef, mean, part, NB, H, W);
  else { bsed_set_error("bsed_conv0_wgrad: first-layer width %d not built (16 or 32)", CO); return BSED_ERR_ARG; }
  BSED_LAUNCH_CHECK();
  return BSED_OK;
}

#define STATS_CHUNKS 128

extern "C" size_t bsed_stats_scratch_bytes(int C) { return (size_t)STATS_CHUNKS * 2 * C * sizeof(double); }

static int stats_reduce_finish(const float* partial, long ntiles, int C, double* scratch, const StatsFin& F, hipStream_t s) {
  const int chunks = (int)std::min<long>(STATS_CHUNKS, ntiles);
  hipLaunchKernelGGL(stats_chunk_kernel, dim3(chunks, ceil_div(C, 32)), dim3(256), 0, s, partial, ntiles, C, scratch);
  hipLaunchKernelGGL(stats_finish_kernel, dim3(ceil_div(C, 32)), dim3(256), 0, s, scratch, chunks, C, F);
  BSED_LAUNCH_CHECK();
  return BSED_OK;
}

extern "C" int bsed_bn_finalize(const float* partial, long ntiles, int C, double count, float eps, float momentum,
                                const float* gamma, const float* beta, float* running_mean, float* running_var,
                                long long* num_batches_tracked, float* mean, float* invstd, float* scale, float* shift,
                                void* scratch, void* stream) {
  BSED_CHECK_ARG(partial && gamma && beta && mean && invstd && scale && shift && scratch, "bsed_bn_finalize: null tensor");
  BSED_CHECK_ARG(ntiles > 0 && C > 0 && count > 0, "bsed_bn_finalize: bad shape");
  hipStream_t s = (hipStream_t)stream;
  double* sc = (double*)scratch;
  StatsFin F = {};
  F.mode = FIN_BN_FWD; F.count = count; F.eps = eps; F.momentum = momentum; F.gamma = gamma; F.beta = beta;
  F.running_mean = running_mean; F.running_var = running_var; F.nbt = num_batches_tracked;
  F.mean_out = mean; F.invstd_out = invstd; F.scale = scale; F.shift = shift;
  return stats_reduce_finish(partial, ntiles, C, sc, F, s);
}

extern "C" int bsed_bn_eval(int C, float eps, const float* gamma, const float* beta, const float* running_mean,
                            const float* running_var, float* scale, float* shift, void* stream) {
  BSED_CHECK_ARG(gamma && beta && running_mean && running_var && scale && shift && C > 0, "bsed_bn_eval: bad argument");
  hipLaunchKernelGGL(bn_eval_kernel, dim3(ceil_div(C, 128)), dim3(128), 0, (hipStream_t)stream, C, eps, gamma, beta,
                     running_mean, running_var, scale, shift);
  BSED_LAUNCH_CHECK();
  return BSED_OK;
}

extern "C" int bsed_bn_eval_batch(const BsedBnEvalJob* jobs, int njobs, float eps, void* stream) {
  BSED_CHECK_ARG(jobs && njobs > 0 && njobs <= BSED_BN_EVAL_MAX_JOBS, "bsed_bn_eval_batch: 1..%d jobs", BSED_BN_EVAL_MAX_JOBS);
  BnEvalBatch Bt;
  Bt.eps = eps;
  int cmax = 0;
  for (int i = 0; i < njobs; ++i) {
    const BsedBnEvalJob& q = jobs[i];
    BSED_CHECK_ARG(q.gamma && q.beta && q.running_mean && q.running_var && q.scale && q.shift && q.C > 0,
                   "bsed_bn_eval_batch: bad job %d", i);
    Bt.j[i].gamma = q.gamma; Bt.j[i].beta = q.beta; Bt.j[i].rmean = q.running_mean; Bt.j[i].rvar = q.running_var;
    Bt.j[i].scale = q.scale; Bt.j[i].shift = q.shift; Bt.j[i].C = q.C;
    cmax = std::max(cmax, q.C);
  }
  hipLaunchKernelGGL(bn_eval_batch_kernel, dim3(ceil_div(cmax, 128), njobs), dim3(128), 0, (hipStream_t)stream, Bt);
  BSED_LAUNCH_CHECK();
  return BSED_OK;
}

extern "C" int bsed_bn_bwd(const float* partial, long ntiles, int C, double count, const float* gamma,
                           const float* mean, const float* invstd, float* dgamma, float* dbeta, int accumulate,
                           float* g_inout, const float* y, long n_elems, float* coef, void* scratch, void* stream) {
  BSED_CHECK_ARG(partial && gamma && mean && invstd && dgamma && dbeta && coef && scratch, "bsed_bn_bwd: null tensor");
  BSED_CHECK_ARG((g_inout == nullptr) == (y == nullptr), "bsed_bn_bwd: g_inout and y come together");
  BSED_CHECK_ARG(ntiles > 0 && C > 0 && C % 4 == 0 && n_elems % C == 0, "bsed_bn_bwd: bad shape");
  hipStream_t s = (hipStream_t)stream;
  double* sc = (double*)scratch;
  StatsFin F = {};
  F.mode = FIN_BN_BWD; F.count = count; F.gamma = gamma; F.mean = mean; F.invstd = invstd;
  F.dgamma = dgamma; F.dbeta = dbeta; F.accumulate = accumulate; F.coef = coef;
  int rc = stats_reduce_finish(partial, ntiles, C, sc, F, s);
  if (rc) return rc;
  if (!g_inout) {  // coefficients only: the consumer applies the affine map on load (bsed_conv0_wgrad)
    BSED_LAUNCH_CHECK();
    return BSED_OK;
  }
  const long n4 = n_elems / 4;
  hipLaunchKernelGGL(bn_bwd_apply_kernel, dim3((unsigned)std::min<long>(ceil_div(n4, 256), 8192)), dim3(256), 0, s,
                     g_inout, y, coef, mean, n4, C);
  BSED_LAUNCH_CHECK();
  return BSED_OK;
}

// dst[c] (+)= sum over tiles of partial[tile][which][c]
extern "C" int bsed_stats_to_grad(const float* partial, long ntiles, int C, int which, float* dst, int accumulate,
                                  void* scratch, void* stream) {
  BSED_CHECK_ARG(partial && dst && scratch && ntiles > 0 && C > 0 && (which == 0 || which == 1), "bsed_stats_to_grad: bad argument");
  hipStream_t s = (hipStream_t)stream;
  double* sc = (double*)scratch;
  StatsFin F = {};
  F.mode = FIN_TO_GRAD; F.which = which; F.dst = dst; F.accumulate = accumulate;
  return stats_reduce_finish(partial, ntiles, C, sc, F, s);
}

// dst[c] (+)= sum_r in[r][c]; part must hold G*2*C floats
extern "C" int bsed_colsum(const float* in, long M, int C, int pitch, float* part, int G, float* dst, int accumulate,
                           void* scratch, void* stream) {
  BSED_CHECK_ARG(in && part && dst && scratch && M > 0 && C > 0 && pitch >= C && G > 0, "bsed_colsum: bad argument");
  hipStream_t s = (hipStream_t)stream;
  const int g = (int)std::min<long>(G, M);
  hipLaunchKernelGGL(colsum_kernel, dim3(g), dim3(256), 0, s, in, M, C, pitch, part);
  return bsed_stats_to_grad(part, g, C, 0, dst, accumulate, scratch, stream);
}

// the first stage alone: part (min(G, M), 2, C) per-workgroup column sums in slot 0, for a queued second stage
// (bsed_reduce_partials_batch)
extern "C" int bsed_colsum_part(const float* in, long M, int C, int pitch, float* part, int G, void* stream) {
  BSED_CHECK_ARG(in && part && M > 0 && C > 0 && pitch >= C && G > 0 && G <= M, "bsed_colsum_part: bad argument");
  hipLaunchKernelGGL(colsum_kernel, dim3(G), dim3(256), 0, (hipStream_t)stream, in, M, C, pitch, part);
  BSED_LAUNCH_CHECK();
  return BSED_OK;
}

extern "C" int bsed_dropout(const float* in, float* out, long n, float p, uint32_t rng_stream, uint64_t seed,
                            void* stream) {
  BSED_CHECK_ARG(in && out && n > 0 && p >= 0.f && p < 1.f, "bsed_dropout: bad argument");
  hipLaunchKernelGGL(dropout_kernel, dim3((unsigned)std::min<long>(ceil_div(n, 256), 8192)), dim3(256), 0,
                     (hipStream_t)stream, in, out, n, p, rng_stream, seed, bsed_seed_add_ptr());
  BSED_LAUNCH_CHECK();
  return BSED_OK;
}
