// Clip-level domain discriminator pieces (reference Clip_Discriminator, src/models/CRNN_GRL.py:16-53, fed through
// the gradient-reverse layer of src/DA/grl.py:12-22 and the BCE of src/DA/cdan_frame.py:89-119).
//
// Layers 2 and 3 (128 -> 64 -> 32 channels: 98 % of the discriminator's multiply-adds) run DIRECTLY on the implicit-GEMM
// kernels through a space-to-depth re-layout (s2d_* below).  Layers 1, 4 and 5 (1, 32 and 16 input channels: too few
// for a 32-channel K chunk) keep the im2col lowering:
//   forward   y = im2col(act) @ W            (bsed_igemm, BSED_EPI_STATS for the BatchNorm statistics)
//   weights   dW = im2col(act)^T @ dY         (bsed_wgrad)
//   data      d_act = col2im(dY @ W^T)        (bsed_igemm + col2im)
// so this file only holds the streaming glue around them:
//   im2col_s2_kernel : gather with BatchNorm-apply + LeakyReLU(0.2) of the previous layer fused into the load
//   col2im_s2_kernel : the adjoint gather, fused with LeakyReLU backward and the BatchNorm-backward sums, or with
//                      the -lambda of the gradient-reverse layer for the first layer
//   disc_head_kernel : BN5 + LeakyReLU + AdaptiveAvgPool2d((2,1)) + Linear(16,1) + sigmoid + BCE, forward and backward
// Spatial layout: the reference permutes the (N,T,256) embedding to an image (256 x T); here the image is kept
// as (H = T, W = 256) -- the embedding's own memory order -- and the 3x3 kernels are read transposed instead.
#include "bsed_common.h"
#include "../../include/bsed.h"
#include <algorithm>

#define LEAKY 0.2f

// col[(n,ho,wo)][(dw*3+dh)*CP + c] = f(act[n][2ho+dh][2wo+dw][c]),  f = leaky(x*scale+shift) or identity
__global__ void im2col_s2_kernel(const float* __restrict__ act, const float* __restrict__ scale,
                                 const float* __restrict__ shift, float* __restrict__ col, int N, int Hi, int Wi, int C,
                                 int CP, int KW, int Ho, int Wo) {
  const int cq = CP / 4;                       // float4 groups per tap (C == 1: handled by the scalar branch)
  const long total = (long)N * Ho * Wo * 9 * cq;
  if (C == 1) {
    const long tot1 = (long)N * Ho * Wo * KW;  // KW = 16: 9 taps + zero padding
    for (long e = (long)blockIdx.x * blockDim.x + threadIdx.x; e < tot1; e += (long)gridDim.x * blockDim.x) {
      const int t = (int)(e % KW);
      const long p = e / KW;
      float v = 0.f;
      if (t < 9) {
        const int dw = t / 3, dh = t % 3;
        const int wo = (int)(p % Wo);
        const long r = p / Wo;
        const int ho = (int)(r % Ho);
        const long n = r / Ho;
        v = act[(n * Hi + 2 * ho + dh) * Wi + 2 * wo + dw];
      }
      col[e] = v;
    }
    return;
  }
  for (long e = (long)blockIdx.x * blockDim.x + threadIdx.x; e < total; e += (long)gridDim.x * blockDim.x) {
    const int c4 = (int)(e % cq);
    long r = e / cq;
    const int t = (int)(r % 9); r /= 9;
    const int wo = (int)(r % Wo); r /= Wo;
    const int ho = (int)(r % Ho);
    const long n = r / Ho;
    const int dw = t / 3, dh = t % 3;
    float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
    const int c = 4 * c4;
    if (c < C) {
      v = *reinterpret_cast<const float4*>(act + ((n * Hi + 2 * ho + dh) * Wi + 2 * wo + dw) * (long)C + c);
      if (scale) {
        const float4 sc = *reinterpret_cast<const float4*>(scale + c);
        const float4 sh = *reinterpret_cast<const float4*>(shift + c);
        v.x = fmaf(v.x, sc.x, sh.x); v.y = fmaf(v.y, sc.y, sh.y);
        v.z = fmaf(v.z, sc.z, sh.z); v.w = fmaf(v.w, sc.w, sh.w);
        v.x = v.x > 0.f ? v.x : LEAKY * v.x; v.y = v.y > 0.f ? v.y : LEAKY * v.y;
        v.z = v.z > 0.f ? v.z : LEAKY * v.z; v.w = v.w > 0.f ? v.w : LEAKY * v.w;
      }
    }
    *reinterpret_cast<float4*>(col + (((n * Ho + ho) * Wo + wo) * 9 + t) * (long)CP + c) = v;
  }
}

// d_act[n][h][w][c] = sum over taps with (h-dh, w-dw) even and in range of dcol[(n,(h-dh)/2,(w-dw)/2)][(dw*3+dh)*CP+c]
//   C > 1 : g = d_act * leaky'(y*scale+shift) written to out, per-block partial sums (sum g, sum g*y) to stats
//   C == 1: out = d_act * out_scale   (gradient-reverse layer: out_scale = -lambda)
__global__ __launch_bounds__(256) void col2im_s2_kernel(const float* __restrict__ dcol, const float* __restrict__ y,
                                                        const float* __restrict__ scale, const float* __restrict__ shift,
                                                        float* __restrict__ out, float* __restrict__ stats, int N, int Hi,
                                                        int Wi, int C, int CP, int KW, int Ho, int Wo, float out_scale) {
  __shared__ float red[256 * 8];
  const int tid = threadIdx.x;
  if (C == 1) {
    const long tot = (long)N * Hi * Wi;
    for (long e = (long)blockIdx.x * blockDim.x + tid; e < tot; e += (long)gridDim.x * blockDim.x) {
      const int w = (int)(e % Wi);
      const long r = e / Wi;
      const int h = (int)(r % Hi);
      const long n = r / Hi;
      float a = 0.f;
      for (int dh = 0; dh < 3; ++dh) {
        const int hh = h - dh;
        if (hh < 0 || (hh & 1) || (hh >> 1) >= Ho) continue;
        for (int dw = 0; dw < 3; ++dw) {
          const int ww = w - dw;
          if (ww < 0 || (ww & 1) || (ww >> 1) >= Wo) continue;
          a += dcol[((n * Ho + (hh >> 1)) * Wo + (ww >> 1)) * (long)KW + dw * 3 + dh];
        }
      }
      out[e] = a * out_scale;
    }
    return;
  }
  const int cq = C / 4;
  const long tot = (long)N * Hi * Wi * cq;
  // one block = 256 consecutive float4 groups; blocks are NOT grid-strided so the partial-stat slot is unique
  const long e = (long)blockIdx.x * 256 + tid;
  float4 g = make_float4(0.f, 0.f, 0.f, 0.f), gy = make_float4(0.f, 0.f, 0.f, 0.f);
  int c = 0;
  if (e < tot) {
    const int c4 = (int)(e % cq);
    c = 4 * c4;
    long r = e / cq;
    const int w = (int)(r % Wi); r /= Wi;
    const int h = (int)(r % Hi);
    const long n = r / Hi;
    float4 a = make_float4(0.f, 0.f, 0.f, 0.f);
    for (int dh = 0; dh < 3; ++dh) {
      const int hh = h - dh;
      if (hh < 0 || (hh & 1) || (hh >> 1) >= Ho) continue;
      for (int dw = 0; dw < 3; ++dw) {
        const int ww = w - dw;
        if (ww < 0 || (ww & 1) || (ww >> 1) >= Wo) continue;
        const float4 v = *reinterpret_cast<const float4*>(
            dcol + (((n * Ho + (hh >> 1)) * Wo + (ww >> 1)) * 9 + dw * 3 + dh) * (long)CP + c);
        a.x += v.x; a.y += v.y; a.z += v.z; a.w += v.w;
      }
    }
    const float4 yv = *reinterpret_cast<const float4*>(y + e * 4);
    const float4 sc = *reinterpret_cast<const float4*>(scale + c);
    const float4 sh = *reinterpret_cast<const float4*>(shift + c);
    g.x = a.x * (fmaf(yv.x, sc.x, sh.x) > 0.f ? 1.f : LEAKY);
    g.y = a.y * (fmaf(yv.y, sc.y, sh.y) > 0.f ? 1.f : LEAKY);
    g.z = a.z * (fmaf(yv.z, sc.z, sh.z) > 0.f ? 1.f : LEAKY);
    g.w = a.w * (fmaf(yv.w, sc.w, sh.w) > 0.f ? 1.f : LEAKY);
    *reinterpret_cast<float4*>(out + e * 4) = g;
    gy = make_float4(g.x * yv.x, g.y * yv.y, g.z * yv.z, g.w * yv.w);
  }
  // per-block sums per channel: thread t holds channel group (e % cq); 256 % cq == 0 for cq in {4,8,16,32}
  float* r8 = red + tid * 8;
  r8[0] = g.x; r8[1] = g.y; r8[2] = g.z; r8[3] = g.w; r8[4] = gy.x; r8[5] = gy.y; r8[6] = gy.z; r8[7] = gy.w;
  __syncthreads();
  if (tid < 2 * C) {
    const int which = tid / C, ch = tid % C;
    const int q0 = (int)(((long)blockIdx.x * 256) % cq);  // channel group of thread 0
    float s = 0.f;
    // threads whose group == ch/4: t with (q0 + t) % cq == ch/4
    int t0 = (ch / 4 - q0) % cq;
    if (t0 < 0) t0 += cq;
    for (int t = t0; t < 256; t += cq) s += red[t * 8 + which * 4 + (ch & 3)];
    stats[((size_t)blockIdx.x * 2 + which) * C + ch] = s;
  }
}

// ---------------------------------------------------------------------------------------------
// Direct form of the stride-2 convolutions (no im2col matrix): space-to-depth.
//   Y[i][j] = sum_{kh,kw,c} W[kh][kw][c] A[2i+kh][2j+kw][c]  with  2i+kh = 2(i + (kh>>1)) + (kh&1):
//   X'[p][q][(a*2+b)*C + c] = A[2p+a][2q+b][c]  turns it into a 2x2 stride-1 convolution over 4C channels whose taps
//   (dp,dq) in {0,1}^2 carry W[2dp+a][2dq+b] (zero where 2dp+a or 2dq+b would be 3).  16/9 of the multiply-adds, but
//   the operand is the activation itself (re-laid out once, with the previous layer's BatchNorm + LeakyReLU fused)
//   instead of a 9x larger gathered copy that four GEMM passes stream through HBM.
// s2d_fwd_kernel : y (N,Ha,Wa,C) allocated, valid extent Hi x Wi -> X' (N,Hp,Wp,4C); one float4 of channels per thread
// s2d_bwd_kernel : dX' -> g over the allocated grid of y (LeakyReLU' applied, zero outside the valid extent) + per-block
//                  (sum g, sum g*y) partials, same slot logic as col2im_s2_kernel
// ---------------------------------------------------------------------------------------------
__global__ void s2d_fwd_kernel(const float* __restrict__ y, const float* __restrict__ scale, const float* __restrict__ shift,
                               float* __restrict__ xp, int N, int Ha, int Wa, int Hi, int Wi, int C, int Hp, int Wp) {
  const int cq = C / 4;
  const long total = (long)N * Hp * Wp * 4 * cq;
  for (long e = (long)blockIdx.x * blockDim.x + threadIdx.x; e < total; e += (long)gridDim.x * blockDim.x) {
    const int c4 = (int)(e % cq);
    long r = e / cq;
    const int ab = (int)(r & 3); r >>= 2;
    const int q = (int)(r % Wp); r /= Wp;
    const int pp = (int)(r % Hp);
    const long n = r / Hp;
    const int h = 2 * pp + (ab >> 1), w = 2 * q + (ab & 1), c = 4 * c4;
    float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
    if (h < Hi && w < Wi) {
      v = *reinterpret_cast<const float4*>(y + ((n * Ha + h) * Wa + w) * (long)C + c);
      if (scale) {
        const float4 sc = *reinterpret_cast<const float4*>(scale + c);
        const float4 sh = *reinterpret_cast<const float4*>(shift + c);
        v.x = fmaf(v.x, sc.x, sh.x); v.y = fmaf(v.y, sc.y, sh.y);
        v.z = fmaf(v.z, sc.z, sh.z); v.w = fmaf(v.w, sc.w, sh.w);
        v.x = v.x > 0.f ? v.x : LEAKY * v.x; v.y = v.y > 0.f ? v.y : LEAKY * v.y;
        v.z = v.z > 0.f ? v.z : LEAKY * v.z; v.w = v.w > 0.f ? v.w : LEAKY * v.w;
      }
    }
    *reinterpret_cast<float4*>(xp + e * 4) = v;   // e enumerates X' in memory order: [n][p][q][a*2+b][c]
  }
}

__global__ __launch_bounds__(256) void s2d_bwd_kernel(const float* __restrict__ dxp, const float* __restrict__ y,
                                                      const float* __restrict__ scale, const float* __restrict__ shift,
                                                      float* __restrict__ g_out, float* __restrict__ stats, int N, int Ha,
                                                      int Wa, int Hi, int Wi, int C, int Hp, int Wp) {
  __shared__ float red[256 * 8];
  const int tid = threadIdx.x;
  const int cq = C / 4;
  const long tot = (long)N * Ha * Wa * cq;
  const long e = (long)blockIdx.x * 256 + tid;   // not grid-strided: the partial-stat slot is the block's own
  float4 g = make_float4(0.f, 0.f, 0.f, 0.f), gy = make_float4(0.f, 0.f, 0.f, 0.f);
  if (e < tot) {
    const int c4 = (int)(e % cq), c = 4 * c4;
    long r = e / cq;
    const int w = (int)(r % Wa); r /= Wa;
    const int h = (int)(r % Ha);
    const long n = r / Ha;
    if (h < Hi && w < Wi) {
      const float4 a = *reinterpret_cast<const float4*>(
          dxp + ((((n * Hp + (h >> 1)) * Wp + (w >> 1)) * 4 + ((h & 1) * 2 + (w & 1))) * (long)C + c));
      const float4 yv = *reinterpret_cast<const float4*>(y + e * 4);
      const float4 sc = *reinterpret_cast<const float4*>(scale + c);
      const float4 sh = *reinterpret_cast<const float4*>(shift + c);
      g.x = a.x * (fmaf(yv.x, sc.x, sh.x) > 0.f ? 1.f : LEAKY);
      g.y = a.y * (fmaf(yv.y, sc.y, sh.y) > 0.f ? 1.f : LEAKY);
      g.z = a.z * (fmaf(yv.z, sc.z, sh.z) > 0.f ? 1.f : LEAKY);
      g.w = a.w * (fmaf(yv.w, sc.w, sh.w) > 0.f ? 1.f : LEAKY);
      gy = make_float4(g.x * yv.x, g.y * yv.y, g.z * yv.z, g.w * yv.w);
    }
    *reinterpret_cast<float4*>(g_out + e * 4) = g;
  }
  float* r8 = red + tid * 8;
  r8[0] = g.x; r8[1] = g.y; r8[2] = g.z; r8[3] = g.w; r8[4] = gy.x; r8[5] = gy.y; r8[6] = gy.z; r8[7] = gy.w;
  __syncthreads();
  if (tid < 2 * C) {
    const int which = tid / C, ch = tid % C;
    const int q0 = (int)(((long)blockIdx.x * 256) % cq);
    float s = 0.f;
    int t0 = (ch / 4 - q0) % cq;
    if (t0 < 0) t0 += cq;
    for (int t = t0; t < 256; t += cq) s += red[t * 8 + which * 4 + (ch & 3)];
    stats[((size_t)blockIdx.x * 2 + which) * C + ch] = s;
  }
}

// tail of the discriminator for one sample per thread: y5 (N,H5,W5,8) -> BN + LeakyReLU -> AdaptiveAvgPool over the
// feature axis (W5 = 7 -> bins [0,4), [3,7)) and all of time -> Linear(16,1) -> sigmoid -> BCE(label_n), and back.
__global__ void disc_head_kernel(const float* __restrict__ y5, const float* __restrict__ scale,
                                 const float* __restrict__ shift, const float* __restrict__ wl, const float* __restrict__ bl,
                                 int N, int Ns, int H5, int W5, int train, float* __restrict__ d_out,
                                 float* __restrict__ g5, float* __restrict__ stats /*(N,2,8)*/,
                                 float* __restrict__ dwl_part /*(N,2,16)*/, float* __restrict__ dbl_part /*(N,2,1)*/,
                                 float* __restrict__ loss_part /*(N,2,1)*/) {
  const int n = blockIdx.x * blockDim.x + threadIdx.x;
  if (n >= N) return;
  constexpr int C = 8;
  const float* yn = y5 + (size_t)n * H5 * W5 * C;
  // adaptive bins over W5 (PyTorch: start = floor(i*W/2), end = ceil((i+1)*W/2))
  const int b0s = 0, b0e = (W5 + 1) / 2, b1s = W5 / 2, b1e = W5;
  float pooled[C][2];
  for (int c = 0; c < C; ++c) { pooled[c][0] = 0.f; pooled[c][1] = 0.f; }
  for (int h = 0; h < H5; ++h)
    for (int w = 0; w < W5; ++w)
      for (int c = 0; c < C; ++c) {
        float a = fmaf(yn[(h * W5 + w) * C + c], scale[c], shift[c]);
        a = a > 0.f ? a : LEAKY * a;
        if (w >= b0s && w < b0e) pooled[c][0] += a;
        if (w >= b1s && w < b1e) pooled[c][1] += a;
      }
  const float inv0 = 1.0f / (float)(H5 * (b0e - b0s)), inv1 = 1.0f / (float)(H5 * (b1e - b1s));
  float z = bl[0];
  for (int c = 0; c < C; ++c) {
    pooled[c][0] *= inv0; pooled[c][1] *= inv1;
    z = fmaf(wl[c * 2], pooled[c][0], z);
    z = fmaf(wl[c * 2 + 1], pooled[c][1], z);
  }
  const float d = sigmoidf_(z);
  d_out[n] = d;
  if (!train) return;
  const float lab = n < Ns ? 1.f : 0.f;
  loss_part[n * 2] = -(lab * fmaxf(logf(d), -100.f) + (1.f - lab) * fmaxf(logf(1.f - d), -100.f));
  loss_part[n * 2 + 1] = 0.f;
  const float dz = (d - lab) / fmaxf((1.f - d) * d, 1e-12f) / (float)N * d * (1.f - d);
  dbl_part[n * 2] = dz; dbl_part[n * 2 + 1] = 0.f;
  float dp[C][2];
  for (int c = 0; c < C; ++c) {
    dwl_part[(size_t)n * 32 + c * 2] = dz * pooled[c][0];
    dwl_part[(size_t)n * 32 + c * 2 + 1] = dz * pooled[c][1];
    dp[c][0] = dz * wl[c * 2] * inv0;
    dp[c][1] = dz * wl[c * 2 + 1] * inv1;
  }
  for (int j = 0; j < 16; ++j) dwl_part[(size_t)n * 32 + 16 + j] = 0.f;
  float sg[C], sgy[C];
  for (int c = 0; c < C; ++c) { sg[c] = 0.f; sgy[c] = 0.f; }
  for (int h = 0; h < H5; ++h)
    for (int w = 0; w < W5; ++w)
      for (int c = 0; c < C; ++c) {
        const float yv = yn[(h * W5 + w) * C + c];
        const float xn = fmaf(yv, scale[c], shift[c]);
        float da = 0.f;
        if (w >= b0s && w < b0e) da += dp[c][0];
        if (w >= b1s && w < b1e) da += dp[c][1];
        const float g = da * (xn > 0.f ? 1.f : LEAKY);
        g5[((size_t)n * H5 * W5 + h * W5 + w) * C + c] = g;
        sg[c] += g;
        sgy[c] = fmaf(g, yv, sgy[c]);
      }
  for (int c = 0; c < C; ++c) {
    stats[((size_t)n * 2 + 0) * C + c] = sg[c];
    stats[((size_t)n * 2 + 1) * C + c] = sgy[c];
  }
}

// ---------------------------------------------------------------------------------------------
// Frame-level discriminator glue (reference Frame_Discriminator, src/models/CRNN_GRL.py:116-140: three per-frame Linear
// layers 256 -> 128 -> 32 -> 1 with LeakyReLU(0.2) + Dropout between them and a sigmoid at the end).  The two wide
// layers are 1-tap contractions on the implicit-GEMM kernels; this file holds the elementwise stages and the 32 -> 1 head.
//   leaky_dropout_fwd/bwd : out = leaky(a) * mask,  d_a = d_out * mask * leaky'(a)   (mask: counter hash, regenerated)
//   frame_head_fwd        : d[m] = sigmoid(sum_k x[m][k] w[k] + b)
//   frame_head_bwd        : dz = d_out * d (1 - d);  dx[m][k] = dz w[k];  per-block partials of dw[k] = sum dz x[m][k], db
// ---------------------------------------------------------------------------------------------
__global__ void leaky_dropout_fwd_kernel(const float* __restrict__ a, float* __restrict__ out, long n4, float slope,
                                         float drop_p, uint32_t rng_stream, uint64_t seed) {
  const uint32_t dkey = drop_key(rng_stream, seed), dthr = drop_threshold(drop_p);
  const float dscale = drop_p > 0.f ? 1.0f / (1.0f - drop_p) : 1.0f;
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += (long)gridDim.x * blockDim.x) {
    const float4 v = reinterpret_cast<const float4*>(a)[i];
    float r[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
    for (int k = 0; k < 4; ++k)
      r[k] = (r[k] > 0.f ? r[k] : slope * r[k]) * (drop_p > 0.f ? drop_mul((uint64_t)(4 * i + k), dkey, dthr, dscale) : 1.f);
    reinterpret_cast<float4*>(out)[i] = make_float4(r[0], r[1], r[2], r[3]);
  }
}

__global__ void leaky_dropout_bwd_kernel(const float* __restrict__ d_out, const float* __restrict__ a,
                                         float* __restrict__ d_a, long n4, float slope, float drop_p, uint32_t rng_stream,
                                         uint64_t seed) {
  const uint32_t dkey = drop_key(rng_stream, seed), dthr = drop_threshold(drop_p);
  const float dscale = drop_p > 0.f ? 1.0f / (1.0f - drop_p) : 1.0f;
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += (long)gridDim.x * blockDim.x) {
    const float4 g = reinterpret_cast<const float4*>(d_out)[i];
    const float4 v = reinterpret_cast<const float4*>(a)[i];
    const float gv[4] = {g.x, g.y, g.z, g.w}, av[4] = {v.x, v.y, v.z, v.w};
    float r[4];
#pragma unroll
    for (int k = 0; k < 4; ++k)
      r[k] = gv[k] * (av[k] > 0.f ? 1.f : slope) * (drop_p > 0.f ? drop_mul((uint64_t)(4 * i + k), dkey, dthr, dscale) : 1.f);
    reinterpret_cast<float4*>(d_a)[i] = make_float4(r[0], r[1], r[2], r[3]);
  }
}

#define FH_K 32
__global__ __launch_bounds__(256) void frame_head_fwd_kernel(const float* __restrict__ x, const float* __restrict__ w,
                                                             const float* __restrict__ b, float* __restrict__ d, long M) {
  const long m = (long)blockIdx.x * 256 + threadIdx.x;
  if (m >= M) return;
  float z = b[0];
#pragma unroll
  for (int k = 0; k < FH_K; k += 4) {
    const float4 v = *reinterpret_cast<const float4*>(x + m * FH_K + k);
    z = fmaf(v.x, w[k], fmaf(v.y, w[k + 1], fmaf(v.z, w[k + 2], fmaf(v.w, w[k + 3], z))));
  }
  d[m] = sigmoidf_(z);
}

__global__ __launch_bounds__(256) void frame_head_bwd_kernel(const float* __restrict__ x, const float* __restrict__ w,
                                                             const float* __restrict__ d, const float* __restrict__ d_out,
                                                             float* __restrict__ dx, float* __restrict__ part /*(G,2,32)*/,
                                                             long M) {
  __shared__ float red[256 * (FH_K + 1)];
  const int tid = threadIdx.x;
  float acc[FH_K], accb = 0.f;
#pragma unroll
  for (int k = 0; k < FH_K; ++k) acc[k] = 0.f;
  for (long m = (long)blockIdx.x * 256 + tid; m < M; m += (long)gridDim.x * 256) {
    const float s = d[m];
    const float dz = d_out[m] * s * (1.f - s);
    accb += dz;
#pragma unroll
    for (int k = 0; k < FH_K; k += 4) {
      const float4 v = *reinterpret_cast<const float4*>(x + m * FH_K + k);
      acc[k] = fmaf(dz, v.x, acc[k]); acc[k + 1] = fmaf(dz, v.y, acc[k + 1]);
      acc[k + 2] = fmaf(dz, v.z, acc[k + 2]); acc[k + 3] = fmaf(dz, v.w, acc[k + 3]);
      *reinterpret_cast<float4*>(dx + m * FH_K + k) = make_float4(dz * w[k], dz * w[k + 1], dz * w[k + 2], dz * w[k + 3]);
    }
  }
#pragma unroll
  for (int k = 0; k < FH_K; ++k) red[tid * (FH_K + 1) + k] = acc[k];
  red[tid * (FH_K + 1) + FH_K] = accb;
  __syncthreads();
  if (tid <= FH_K) {
    float s = 0.f;
    for (int t = 0; t < 256; ++t) s += red[t * (FH_K + 1) + tid];   // fixed order
    if (tid < FH_K) { part[((size_t)blockIdx.x * 2 + 0) * FH_K + tid] = s; part[((size_t)blockIdx.x * 2 + 1) * FH_K + tid] = 0.f; }
    else part[((size_t)blockIdx.x * 2 + 1) * FH_K] = s;             // db rides in row 1, column 0
  }
}

extern "C" int bsed_leaky_dropout_fwd(const float* a, float* out, long n, float slope, float drop_p, uint32_t rng_stream,
                                      uint64_t seed, void* stream) {
  BSED_CHECK_ARG(a && out && n > 0 && n % 4 == 0 && drop_p >= 0.f && drop_p < 1.f, "bsed_leaky_dropout_fwd: bad argument");
  hipLaunchKernelGGL(leaky_dropout_fwd_kernel, dim3((unsigned)std::min<long>(ceil_div(n / 4, 256), 16384)), dim3(256), 0,
                     (hipStream_t)stream, a, out, n / 4, slope, drop_p, rng_stream, seed);
  BSED_LAUNCH_CHECK();
  return BSED_OK;
}

extern "C" int bsed_leaky_dropout_bwd(const float* d_out, const float* a, float* d_a, long n, float slope, float drop_p,
                                      uint32_t rng_stream, uint64_t seed, void* stream) {
  BSED_CHECK_ARG(d_out && a && d_a && n > 0 && n % 4 == 0 && drop_p >= 0.f && drop_p < 1.f, "bsed_leaky_dropout_bwd: bad argument");
  hipLaunchKernelGGL(leaky_dropout_bwd_kernel, dim3((unsigned)std::min<long>(ceil_div(n / 4, 256), 16384)), dim3(256), 0,
                     (hipStream_t)stream, d_out, a, d_a, n / 4, slope, drop_p, rng_stream, seed);
  BSED_LAUNCH_CHECK();
  return BSED_OK;
}

extern "C" int bsed_frame_head_fwd(const float* x, const float* w, const float* b, float* d, long M, int K, void* stream) {
  BSED_CHECK_ARG(x && w && b && d && M > 0, "bsed_frame_head_fwd: bad argument");
  BSED_CHECK_ARG(K == FH_K, "bsed_frame_head_fwd: built for K = %d (got %d)", FH_K, K);
  hipLaunchKernelGGL(frame_head_fwd_kernel, dim3((unsigned)ceil_div(M, 256)), dim3(256), 0, (hipStream_t)stream, x, w, b, d, M);
  BSED_LAUNCH_CHECK();
  return BSED_OK;
}

extern "C" int bsed_frame_head_bwd(const float* x, const float* w, const float* d, const float* d_out, float* dx,
                                   float* part, int G, long M, int K, void* stream) {
  BSED_CHECK_ARG(x && w && d && d_out && dx && part && M > 0 && G > 0 && G <= 4096, "bsed_frame_head_bwd: bad argument");
  BSED_CHECK_ARG(K == FH_K, "bsed_frame_head_bwd: built for K = %d (got %d)", FH_K, K);
  hipLaunchKernelGGL(frame_head_bwd_kernel, dim3(G), dim3(256), 0, (hipStream_t)stream, x, w, d, d_out, dx, part, M);
  BSED_LAUNCH_CHECK();
  return BSED_OK;
}

extern "C" int bsed_im2col_s2(const float* act, const float* scale, const float* shift, float* col, int N, int Hi, int Wi,
                              int C, int CP, void* stream) {
  BSED_CHECK_ARG(act && col && N > 0 && Hi >= 3 && Wi >= 3, "bsed_im2col_s2: bad argument");
  BSED_CHECK_ARG(C == 1 || (C % 4 == 0 && CP >= C && CP % 4 == 0), "bsed_im2col_s2: C must be 1 or a multiple of 4");
  BSED_CHECK_ARG(C == 1 || ((scale == nullptr) == (shift == nullptr)), "bsed_im2col_s2: scale/shift come together");
  const int Ho = (Hi - 3) / 2 + 1, Wo = (Wi - 3) / 2 + 1;
  const int KW = C == 1 ? 16 : 9 * CP;
  const long total = C == 1 ? (long)N * Ho * Wo * KW : (long)N * Ho * Wo * 9 * (CP / 4);
  hipLaunchKernelGGL(im2col_s2_kernel, dim3((unsigned)std::min<long>(ceil_div(total, 256), 16384)), dim3(256), 0,
                     (hipStream_t)stream, act, scale, shift, col, N, Hi, Wi, C, CP, KW, Ho, Wo);
  BSED_LAUNCH_CHECK();
  return BSED_OK;
}

extern "C" int bsed_col2im_s2_num_blocks(int N, int Hi, int Wi, int C) {
  return C == 1 ? 0 : ceil_div((long)N * Hi * Wi * (C / 4), 256);
}

extern "C" int bsed_col2im_s2(const float* dcol, const float* y, const float* scale, const float* shift, float* out,
                              float* stats, int N, int Hi, int Wi, int C, int CP, float out_scale, void* stream) {
  BSED_CHECK_ARG(dcol && out && N > 0 && Hi >= 3 && Wi >= 3, "bsed_col2im_s2: bad argument");
  const int Ho = (Hi - 3) / 2 + 1, Wo = (Wi - 3) / 2 + 1;
  hipStream_t s = (hipStream_t)stream;
  if (C == 1) {
    const long tot = (long)N * Hi * Wi;
    hipLaunchKernelGGL(col2im_s2_kernel, dim3((unsigned)std::min<long>(ceil_div(tot, 256), 16384)), dim3(256), 0, s, dcol,
                       y, scale, shift, out, stats, N, Hi, Wi, C, CP, 16, Ho, Wo, out_scale);
  } else {
    BSED_CHECK_ARG(y && scale && shift && stats, "bsed_col2im_s2: y/scale/shift/stats needed for C > 1");
    BSED_CHECK_ARG(C % 4 == 0 && 256 % (C / 4) == 0 && C <= 128 && CP >= C, "bsed_col2im_s2: C must be 16..128, power of two");
    const long blocks = ceil_div((long)N * Hi * Wi * (C / 4), 256);
    BSED_CHECK_ARG(blocks < (1L << 31), "bsed_col2im_s2: too many blocks");
    hipLaunchKernelGGL(col2im_s2_kernel, dim3((unsigned)blocks), dim3(256), 0, s, dcol, y, scale, shift, out, stats, N, Hi,
                       Wi, C, CP, 9 * CP, Ho, Wo, out_scale);
  }
  BSED_LAUNCH_CHECK();
  return BSED_OK;
}

extern "C" int bsed_s2d_fwd(const float* y, const float* scale, const float* shift, float* xp, int N, int Ha, int Wa, int Hi,
                            int Wi, int C, void* stream) {
  BSED_CHECK_ARG(y && xp && N > 0 && Hi >= 3 && Wi >= 3 && Ha >= Hi && Wa >= Wi, "bsed_s2d_fwd: bad argument");
  BSED_CHECK_ARG(C % 4 == 0 && C >= 4, "bsed_s2d_fwd: C must be a multiple of 4");
  BSED_CHECK_ARG((scale == nullptr) == (shift == nullptr), "bsed_s2d_fwd: scale/shift come together");
  const int Hp = (Hi + 1) / 2, Wp = (Wi + 1) / 2;
  const long total = (long)N * Hp * Wp * C;
  hipLaunchKernelGGL(s2d_fwd_kernel, dim3((unsigned)std::min<long>(ceil_div(total, 256), 32768)), dim3(256), 0,
                     (hipStream_t)stream, y, scale, shift, xp, N, Ha, Wa, Hi, Wi, C, Hp, Wp);
  BSED_LAUNCH_CHECK();
  return BSED_OK;
}

extern "C" int bsed_s2d_num_blocks(int N, int Ha, int Wa, int C) { return ceil_div((long)N * Ha * Wa * (C / 4), 256); }

extern "C" int bsed_s2d_bwd(const float* dxp, const float* y, const float* scale, const float* shift, float* g, float* stats,
                            int N, int Ha, int Wa, int Hi, int Wi, int C, void* stream) {
  BSED_CHECK_ARG(dxp && y && scale && shift && g && stats && N > 0 && Ha >= Hi && Wa >= Wi && Hi >= 3 && Wi >= 3,
                 "bsed_s2d_bwd: bad argument");
  BSED_CHECK_ARG(C % 4 == 0 && 256 % (C / 4) == 0 && C <= 128, "bsed_s2d_bwd: C must be 16..128, a power of two");
  const int Hp = (Hi + 1) / 2, Wp = (Wi + 1) / 2;
  const long blocks = ceil_div((long)N * Ha * Wa * (C / 4), 256);
  BSED_CHECK_ARG(blocks < (1L << 31), "bsed_s2d_bwd: too many blocks");
  hipLaunchKernelGGL(s2d_bwd_kernel, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, dxp, y, scale, shift, g,
                     stats, N, Ha, Wa, Hi, Wi, C, Hp, Wp);
  BSED_LAUNCH_CHECK();
  return BSED_OK;
}

extern "C" int bsed_disc_head(const float* y5, const float* scale, const float* shift, const float* wl, const float* bl,
                              int N, int Ns, int H5, int W5, int C5, int train, float* d_out, float* g5, float* stats,
                              float* dwl_part, float* dbl_part, float* loss_part, void* stream) {
  BSED_CHECK_ARG(y5 && scale && shift && wl && bl && d_out && N > 0 && Ns >= 0 && Ns <= N, "bsed_disc_head: bad argument");
  BSED_CHECK_ARG(C5 == 8 && H5 > 0 && W5 >= 2, "bsed_disc_head: built for the 8-channel last layer");
  BSED_CHECK_ARG(!train || (g5 && stats && dwl_part && dbl_part && loss_part), "bsed_disc_head: training needs the gradient buffers");
  hipLaunchKernelGGL(disc_head_kernel, dim3(ceil_div(N, 64)), dim3(64), 0, (hipStream_t)stream, y5, scale, shift, wl, bl,
                     N, Ns, H5, W5, train, d_out, g5, stats, dwl_part, dbl_part, loss_part);
  BSED_LAUNCH_CHECK();
  return BSED_OK;
}
