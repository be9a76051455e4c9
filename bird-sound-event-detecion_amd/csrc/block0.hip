// The FIRST CNN block (conv3x3 1 -> 16, BatchNorm, GLU, dropout, avg-pool; src/models/CNN.py:46-67 with i = 0) without
// its two largest tensors.  The block's conv output y0 (B, H, 128, 16) -- 1.8 GB at B = 256 -- and its gradient twin
// were the largest tensors of the network; written once and read three times per step they cost ~9 GB of HBM traffic
// for a layer whose input is 113 MB.  Cin = 1 makes y0 a 9-tap stencil of the dB-mel map (144 FMA per position), cheaper
// to RECOMPUTE than to move, and the two things that seemed to need the stored tensors do not:
//
//   * train-mode BatchNorm needs the batch statistics of y0 before anything can be normalised: b0_stats_kernel computes
//     sum y / sum y^2 per channel from x alone (same FMA chain as the consumers, so these ARE the statistics of the values
//     the consumers see), never writing y0;
//   * conv0's weight gradient needs d_y = A g + B (y - mean) + C (BatchNorm backward), whose coefficients depend on
//     whole-batch sums of g: but A, B, C are per-channel constants, so
//         dW[c][t] = sum_pos d_y[pos][c] x_t[pos]
//                  = A_c Gx[c][t] + B_c (Yx[c][t] - mean_c Sx[t]) + C_c Sx[t],
//         Gx[c][t] = sum g x_t   (accumulated by the backward kernel while g is in registers),
//         Yx[c][t] = sum y x_t = sum_t' W[c][t'] R[t'][t] + b_c Sx[t],   R[t'][t] = sum x_t' x_t,  Sx[t] = sum x_t
//     (x_t = the input shifted by tap t, zero padded): R and Sx are 54 numbers of x alone, taken by b0_stats_kernel.
//     g is never written.
//
// Kernels: b0_stats_kernel (x -> BN partials + R / Sx partials), b0_fwd_kernel<PH> (x -> pooled), b0_bwd_kernel<PH>
// (x, d_pooled -> partials of dW_glu, db_glu, BatchNorm-backward sums and Gx), b0_wgrad_finish_kernel (the formula
// above in fp64).  The GLU arithmetic and the lane map are those of glu_small.hip (lane = 16 q + p: column p, channel
// quarter q; both 16x16 contractions on v_mfma_f32_16x16x4_f32, exact fp32).
#include "bsed_common.h"
#include "../../include/bsed.h"
#include <algorithm>

#define B0_C 16
#define B0_NXR 54   // Sx[9] then R packed upper triangle (t <= t'): 45

struct B0W { float w[4][9]; float b[4]; };   // conv taps and bias of the lane's four channels

__device__ __forceinline__ void b0_load_w(const float* __restrict__ cw, const float* __restrict__ cb, int q, B0W& K) {
#pragma unroll
  for (int i = 0; i < 4; ++i) {
#pragma unroll
    for (int t = 0; t < 9; ++t) K.w[i][t] = cw[(4 * q + i) * 9 + t];
    K.b[i] = cb[4 * q + i];
  }
}

// THE definition of y0: bias, then taps 0..8 in order, one fma each (conv0_fwd_kernel uses the same chain)
__device__ __forceinline__ float b0_conv1(const float (&x9)[9], const float (&w)[9], float b) {
  float a = b;
#pragma unroll
  for (int t = 0; t < 9; ++t) a = fmaf(x9[t], w[t], a);
  return a;
}

__device__ float b0_sink[256];

__device__ __forceinline__ f32x4 b0_mm16(const float (&a)[4], float b0, float b1, float b2, float b3, f32x4 c) {
  asm volatile("s_nop 4" : "+v"(b0), "+v"(b1), "+v"(b2), "+v"(b3), "+v"(c));
  c = __builtin_amdgcn_mfma_f32_16x16x4f32(a[0], b0, c, 0, 0, 0);
  c = __builtin_amdgcn_mfma_f32_16x16x4f32(a[1], b1, c, 0, 0, 0);
  c = __builtin_amdgcn_mfma_f32_16x16x4f32(a[2], b2, c, 0, 0, 0);
  c = __builtin_amdgcn_mfma_f32_16x16x4f32(a[3], b3, c, 0, 0, 0);
  return c;
}

// rows r0-1 .. r0+PH of image b at columns w-1, w, w+1 (zero outside the map): the inputs of PH stacked positions
template <int PH>
struct B0X { float v[PH + 2][3]; };

template <int PH>
__device__ __forceinline__ void b0_fetch(const float* __restrict__ x, int b, int r0, int w, int H, int W, B0X<PH>& X) {
  const float* xi = x + (size_t)b * H * W;
#pragma unroll
  for (int j = 0; j < PH + 2; ++j) {
    const int r = r0 - 1 + j;
    const bool rv = r >= 0 && r < H;
    const int rc = min(max(r, 0), H - 1);
#pragma unroll
    for (int k = 0; k < 3; ++k) {
      const int c = w - 1 + k;
      const bool cv = c >= 0 && c < W;
      const int cc = min(max(c, 0), W - 1);
      const float v = xi[(size_t)rc * W + cc];
      X.v[j][k] = (rv && cv) ? v : 0.f;
    }
  }
}

// (image, pooled row, column chunk) of a grid-stride work list without a division per item: the stride is decomposed
// once, then every step is a mixed-radix addition with carries (all wave-uniform: scalar unit)
struct B0Idx {
  int b, hp, ch;      // current item
  int sb, shp, sch;   // gridDim.x in the same radix
  int Hp, chunks;
  __device__ __forceinline__ void init(unsigned first, unsigned stride, int Hp_, int chunks_) {
    Hp = Hp_; chunks = chunks_;
    ch = (int)(first % (unsigned)chunks);
    const unsigned r = first / (unsigned)chunks;
    hp = (int)(r % (unsigned)Hp); b = (int)(r / (unsigned)Hp);
    sch = (int)(stride % (unsigned)chunks);
    const unsigned sr = stride / (unsigned)chunks;
    shp = (int)(sr % (unsigned)Hp); sb = (int)(sr / (unsigned)Hp);
  }
  __device__ __forceinline__ B0Idx next() const {
    B0Idx n = *this;
    n.ch = ch + sch;
    const int c0 = n.ch >= chunks ? 1 : 0;
    n.ch -= c0 ? chunks : 0;
    n.hp = hp + shp + c0;
    const int c1 = n.hp >= Hp ? 1 : 0;
    n.hp -= c1 ? Hp : 0;
    n.b = b + sb + c1;
    return n;
  }
};

// ---------------------------------------------------------------------------------------------
// statistics: one position per thread, persistent over (image, 2-row strips); 16 + 16 + 54 thread-private sums
// ---------------------------------------------------------------------------------------------
#define B0S_THREADS 256
__global__ __launch_bounds__(B0S_THREADS) void b0_stats_kernel(const float* __restrict__ x, const float* __restrict__ cw,
                                                               const float* __restrict__ cb, float* __restrict__ stats,
                                                               float* __restrict__ xr, int NB, int H, int W) {
  __shared__ float red[B0S_THREADS * 9];
  __shared__ float part8[72];
  const int tid = threadIdx.x;
  float sy[B0_C], sq[B0_C], sr[B0_NXR];
#pragma unroll
  for (int c = 0; c < B0_C; ++c) { sy[c] = 0.f; sq[c] = 0.f; }
#pragma unroll
  for (int i = 0; i < B0_NXR; ++i) sr[i] = 0.f;
  // a wave takes 64 consecutive columns of one row: wpr waves per row, 4 / wpr rows per workgroup step (W <= 256)
  const int wave = tid >> 6, lane = tid & 63;
  const int wpr = (W + 63) >> 6, rpw = 4 / wpr;
  const int wr = wave / wpr, w = (wave - wr * wpr) * 64 + lane;
  const int nrows = NB * H;
  for (int r0 = blockIdx.x * rpw; r0 < nrows; r0 += gridDim.x * rpw) {
    // the 160 wave-uniform weights are re-read through the scalar cache every iteration (hoisted out of the loop they
    // exceed the SGPR file and come back as 156 v_readlane per position)
    asm volatile("" ::: "memory");
    const int row = r0 + wr;
    if (row >= nrows || wr >= rpw || w >= W) continue;
    const int nb = row / H, h = row - nb * H;
    B0X<1> X;
    b0_fetch<1>(x, nb, h, w, H, W, X);
    float x9[9];
#pragma unroll
    for (int t = 0; t < 9; ++t) x9[t] = X.v[t / 3][t % 3];
#pragma unroll
    for (int c = 0; c < B0_C; ++c) {
      float wv[9];
#pragma unroll
      for (int t = 0; t < 9; ++t) wv[t] = cw[c * 9 + t];   // wave-uniform: scalar loads, SGPR operands
      const float y = b0_conv1(x9, wv, cb[c]);
      sy[c] += y;
      sq[c] = fmaf(y, y, sq[c]);
    }
    int k = 9;
#pragma unroll
    for (int t = 0; t < 9; ++t) {
      sr[t] += x9[t];
#pragma unroll
      for (int u = t; u < 9; ++u) { sr[k] = fmaf(x9[t], x9[u], sr[k]); ++k; }
    }
  }
  // workgroup sums in a fixed order, 9 of the 86 values per pass: [sum y | sum y^2] -> stats, [Sx | R] -> xr
  float* st = stats + (size_t)blockIdx.x * 2 * B0_C;
  float* xo = xr + (size_t)blockIdx.x * B0_NXR;
#pragma unroll
  for (int pass = 0; pass < 10; ++pass) {
    __syncthreads();   // the previous pass has been read
#pragma unroll
    for (int j = 0; j < 9; ++j) {
      const int e = pass * 9 + j;   // a constant once both loops are unrolled
      red[tid * 9 + j] = e < B0_C ? sy[e < B0_C ? e : 0]
                         : e < 2 * B0_C ? sq[e < 2 * B0_C ? e - B0_C : 0]
                         : e < 2 * B0_C + B0_NXR ? sr[e < 2 * B0_C + B0_NXR ? e - 2 * B0_C : 0] : 0.f;
    }
    __syncthreads();
    if (tid < 72) {   // 8 partial sums of 32 threads per value
      const int j = tid % 9, g = tid / 9;
      float s = 0.f;
      for (int i = 0; i < 32; ++i) s += red[(g * 32 + i) * 9 + j];
      part8[tid] = s;
    }
    __syncthreads();
    if (tid < 9) {
      float s = 0.f;
      for (int g = 0; g < 8; ++g) s += part8[g * 9 + tid];
      const int e = pass * 9 + tid;
      if (e < 2 * B0_C) st[e] = s;
      else if (e < 2 * B0_C + B0_NXR) xo[e - 2 * B0_C] = s;
    }
  }
}

// sums of (G, N) fp32 partial rows in fp64, fixed order: out[n] = sum_g part[g][n]
__global__ __launch_bounds__(256) void b0_colsum64_kernel(const float* __restrict__ part, int G, int N,
                                                          double* __restrict__ out) {
  __shared__ double sm[256];
  const int tid = threadIdx.x, cl = tid & 3, g = tid >> 2;
  const int n = blockIdx.x * 4 + cl;
  double s = 0.0;
  if (n < N)
    for (int r = g; r < G; r += 64) s += (double)part[(size_t)r * N + n];
  sm[tid] = s;
  __syncthreads();
  if (g == 0 && n < N) {
    double t = 0.0;
    for (int k = 0; k < 64; ++k) t += sm[k * 4 + cl];
    out[n] = t;
  }
}

// ---------------------------------------------------------------------------------------------
// forward: x -> conv -> BN-apply -> Linear -> gate -> dropout -> PH x pw average pool -> pooled
// ---------------------------------------------------------------------------------------------
#define B0F_THREADS 512
template <int PH>
__global__ __launch_bounds__(B0F_THREADS) void b0_fwd_kernel(
    const float* __restrict__ x, const float* __restrict__ cw, const float* __restrict__ cb,
    const float* __restrict__ scale, const float* __restrict__ shift, const float* __restrict__ wg,
    const float* __restrict__ bg, float* __restrict__ out, int B, int H, int W, int pw, float drop_p,
    uint32_t rng_stream, uint64_t seed) {
  constexpr int C = B0_C;
  const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63, p = lane & 15, q = lane >> 4;
  const int col = wave * 16 + p;
  const int Hp = H / PH, Wp = W / pw;
  B0W K;
  b0_load_w(cw, cb, q, K);
  float a1[4];
#pragma unroll
  for (int kk = 0; kk < 4; ++kk) a1[kk] = wg[p * C + 4 * q + kk];
  float sc[4], sh[4], bi[4];
#pragma unroll
  for (int i = 0; i < 4; ++i) { sc[i] = scale[4 * q + i]; sh[i] = shift[4 * q + i]; bi[i] = bg[4 * q + i]; }
  const uint32_t dkey = drop_key(rng_stream, seed), dthr = drop_threshold(drop_p);
  const float dscale = drop_p > 0.f ? 1.0f / (1.0f - drop_p) : 1.0f;
  const float inv = 1.0f / (float)(PH * pw);
  const int chunks = (W + B0F_THREADS / 4 - 1) / (B0F_THREADS / 4);
  B0Idx cur, nxt;
  cur.init(blockIdx.x, gridDim.x, Hp, chunks);

  B0X<PH> nx;
  auto fetch = [&](const B0Idx& I) {
    b0_fetch<PH>(x, I.b, I.hp * PH, min(I.ch * (B0F_THREADS / 4) + col, W - 1), H, W, nx);
  };
  fetch(cur);
  for (; cur.b < B; cur = nxt) {
    const B0X<PH> X = nx;
    nxt = cur.next();
    fetch(nxt.b < B ? nxt : cur);   // the last prefetch re-reads the current item: no branch around the loads
    const int hp = cur.hp, b = cur.b;
    const int w = cur.ch * (B0F_THREADS / 4) + col;
    const bool ok = w < W;
    const int wc = ok ? w : W - 1;
    float pooled[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int dh = 0; dh < PH; ++dh) {
      const int h = hp * PH + dh;
      const size_t pos = ((size_t)b * H + h) * W + wc;
      float x9[9];
#pragma unroll
      for (int t = 0; t < 9; ++t) x9[t] = X.v[dh + t / 3][t % 3];
      float xn[4];
#pragma unroll
      for (int i = 0; i < 4; ++i) xn[i] = fmaf(b0_conv1(x9, K.w[i], K.b[i]), sc[i], sh[i]);
      const f32x4 lin = b0_mm16(a1, xn[0], xn[1], xn[2], xn[3], f32x4{0.f, 0.f, 0.f, 0.f});
#pragma unroll
      for (int i = 0; i < 4; ++i)
        pooled[i] += (lin[i] + bi[i]) * sigmoid_fast(xn[i]) *
                     drop_mul((uint64_t)pos * C + 4 * q + i, dkey, dthr, dscale);
    }
    if (pw == 2) {
#pragma unroll
      for (int i = 0; i < 4; ++i) pooled[i] += __shfl_xor(pooled[i], 1, 64);   // the neighbouring column
    }
    if (ok && (w & (pw - 1)) == 0 && (w / pw) < Wp) {
      const float4 o = make_float4(pooled[0] * inv, pooled[1] * inv, pooled[2] * inv, pooled[3] * inv);
      *reinterpret_cast<float4*>(out + (((size_t)b * Hp + hp) * Wp + w / pw) * C + 4 * q) = o;
    }
  }
}

// ---------------------------------------------------------------------------------------------
// backward: x, d_pooled -> partials of dW_glu (G,16,16), db_glu (G,2,16), (sum g, sum g y) (G,2,16), Gx (G,9,16)
// ---------------------------------------------------------------------------------------------
#define B0B_THREADS 256
struct f4b { float v[4]; };
__device__ __forceinline__ f4b b0_shfl_xor4(const f4b& a, int s) {
  f4b r;
#pragma unroll
  for (int i = 0; i < 4; ++i) r.v[i] = __shfl_xor(a.v[i], s, 64);
  return r;
}

template <int PH>
__global__ __launch_bounds__(B0B_THREADS, 2) void b0_bwd_kernel(
    const float* __restrict__ x, const float* __restrict__ cw, const float* __restrict__ cb,
    const float* __restrict__ scale, const float* __restrict__ shift, const float* __restrict__ wg,
    const float* __restrict__ bg, const float* __restrict__ dpool, float* __restrict__ part_dw,
    float* __restrict__ part_db, float* __restrict__ part_st, float* __restrict__ part_gx, int B, int H, int W, int pw,
    float drop_p, uint32_t rng_stream, uint64_t seed) {
  constexpr int C = B0_C;
  __shared__ float red[B0B_THREADS * 17];
  const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63, p = lane & 15, q = lane >> 4;
  const int col = wave * 16 + p;
  const int Hp = H / PH, Wp = W / pw;
  const int spw = pw >> 1;
  B0W K;
  b0_load_w(cw, cb, q, K);
  float a1[4], a2[4];   // W[n = p][4q + kk]  and  W[4q + kk][c = p]
#pragma unroll
  for (int kk = 0; kk < 4; ++kk) { a1[kk] = wg[p * C + 4 * q + kk]; a2[kk] = wg[(4 * q + kk) * C + p]; }
  float sc[4], sh[4], bi[4];
#pragma unroll
  for (int i = 0; i < 4; ++i) { sc[i] = scale[4 * q + i]; sh[i] = shift[4 * q + i]; bi[i] = bg[4 * q + i]; }
  const uint32_t dkey = drop_key(rng_stream, seed), dthr = drop_threshold(drop_p);
  const float dscale = drop_p > 0.f ? 1.0f / (1.0f - drop_p) : 1.0f;
  const float inv = 1.0f / (float)(PH * pw);

  float dwa[4][4][4];   // dW[4q+i][4(q^s)+kk]
  float gxa[4][9];      // Gx[4q+i][t]
  float dba[4] = {0.f, 0.f, 0.f, 0.f}, sga[4] = {0.f, 0.f, 0.f, 0.f}, sgya[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
  for (int s = 0; s < 4; ++s)
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
      for (int kk = 0; kk < 4; ++kk) dwa[s][i][kk] = 0.f;
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int t = 0; t < 9; ++t) gxa[i][t] = 0.f;

  // work list: (image, pooled row, 64-column chunk); rows beyond Hp * PH have zero gradient and are not visited
  const int chunks = (W + B0B_THREADS / 4 - 1) / (B0B_THREADS / 4);
  B0Idx cur, nxt;
  cur.init(blockIdx.x, gridDim.x, Hp, chunks);
  B0X<PH> nx;
  float4 nd = make_float4(0.f, 0.f, 0.f, 0.f);
  auto fetch = [&](const B0Idx& I) {
    const int w = min(I.ch * (B0B_THREADS / 4) + col, W - 1);
    b0_fetch<PH>(x, I.b, I.hp * PH, w, H, W, nx);
    const int wpi = min(w >> spw, Wp - 1);
    nd = *reinterpret_cast<const float4*>(dpool + (((size_t)I.b * Hp + I.hp) * Wp + wpi) * C + 4 * q);
  };
  fetch(cur);
  for (; cur.b < B; cur = nxt) {
    const B0X<PH> X = nx;
    const float4 cd = nd;
    nxt = cur.next();
    fetch(nxt.b < B ? nxt : cur);
    const int hp = cur.hp, b = cur.b;
    const int w = cur.ch * (B0B_THREADS / 4) + col;
    const bool ok = w < W;
    const float pmask = (ok && (w >> spw) < Wp) ? inv : 0.f;
    const float dres[4] = {cd.x * pmask, cd.y * pmask, cd.z * pmask, cd.w * pmask};
#pragma unroll
    for (int dh = 0; dh < PH; ++dh) {
      const int h = hp * PH + dh;
      const size_t pos = ((size_t)b * H + h) * W + min(w, W - 1);
      float x9[9];
#pragma unroll
      for (int t = 0; t < 9; ++t) x9[t] = X.v[dh + t / 3][t % 3];
      float yv[4];
      f4b xs[4];
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        yv[i] = b0_conv1(x9, K.w[i], K.b[i]);
        xs[0].v[i] = fmaf(yv[i], sc[i], sh[i]);
      }
      const f32x4 lin = b0_mm16(a1, xs[0].v[0], xs[0].v[1], xs[0].v[2], xs[0].v[3], f32x4{0.f, 0.f, 0.f, 0.f});
      float dl[4];
      f32x4 gt;
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const float sg = sigmoid_fast(xs[0].v[i]);
        const float dr = dres[i] * drop_mul((uint64_t)pos * C + 4 * q + i, dkey, dthr, dscale);
        dl[i] = dr * sg;
        gt[i] = dr * (lin[i] + bi[i]) * sg * (1.0f - sg);
      }
      const f32x4 g = b0_mm16(a2, dl[0], dl[1], dl[2], dl[3], gt);   // g = d_lin W + gate term (0 on idle columns)
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        dba[i] += dl[i];
        sga[i] += g[i];
        sgya[i] = fmaf(g[i], yv[i], sgya[i]);
#pragma unroll
        for (int t = 0; t < 9; ++t) gxa[i][t] = fmaf(g[i], x9[t], gxa[i][t]);
      }
#pragma unroll
      for (int s = 1; s < 4; ++s) xs[s] = b0_shfl_xor4(xs[0], 16 * s);
#pragma unroll
      for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int s = 0; s < 4; ++s)
#pragma unroll
          for (int kk = 0; kk < 4; ++kk) dwa[s][i][kk] = fmaf(dl[i], xs[s].v[kk], dwa[s][i][kk]);
    }
  }

  // workgroup reduction over the 64 threads that share a quarter index q, 16 values per pass:
  //   passes 0-3 dW (partner quarter q ^ pass), 4 db / sum g / sum g y, 5-7 Gx (12 values each)
#pragma unroll
  for (int pass = 0; pass < 8; ++pass) {
    __syncthreads();
#pragma unroll
    for (int j = 0; j < 16; ++j) {
      float v;
      if (pass < 4) v = dwa[pass][j >> 2][j & 3];
      else if (pass == 4) v = j < 4 ? dba[j & 3] : (j < 8 ? sga[j & 3] : (j < 12 ? sgya[j & 3] : 0.f));
      else {
        const int e = (pass - 5) * 12 + j;   // e = i * 9 + t, 36 values over three passes of 12
        v = 0.f;
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
          for (int t = 0; t < 9; ++t)
            if (j < 12 && e == i * 9 + t) v = gxa[i][t];
      }
      red[tid * 17 + j] = v;
    }
    __syncthreads();
    if (tid < 64) {
      const int rq = tid & 3, j = tid >> 2;
      float s = 0.f;
      for (int wv = 0; wv < B0B_THREADS / 64; ++wv)
        for (int pp = 0; pp < 16; ++pp) s += red[(wv * 64 + 16 * rq + pp) * 17 + j];
      const size_t gblk = blockIdx.x;
      if (pass < 4) {
        part_dw[(gblk * C + 4 * rq + (j >> 2)) * C + 4 * (rq ^ pass) + (j & 3)] = s;
      } else if (pass == 4) {
        if (j < 4) {
          part_db[(gblk * 2 + 0) * C + 4 * rq + j] = s;
          part_db[(gblk * 2 + 1) * C + 4 * rq + j] = 0.f;
        } else if (j < 8) {
          part_st[(gblk * 2 + 0) * C + 4 * rq + (j - 4)] = s;
        } else if (j < 12) {
          part_st[(gblk * 2 + 1) * C + 4 * rq + (j - 8)] = s;
        }
      } else if (j < 12) {
        const int e = (pass - 5) * 12 + j, i = e / 9, t = e % 9;
        part_gx[(gblk * 9 + t) * C + 4 * rq + i] = s;
      }
    }
  }
}

// dW0[c][t] (+)= A_c Gx[c][t] + B_c (Yx[c][t] - mean_c Sx[t]) + C_c Sx[t]  in fp64; one workgroup per 4 (t, c) columns
// of the (G, 9*16) Gx partials
__global__ __launch_bounds__(256) void b0_wgrad_finish_kernel(const float* __restrict__ part_gx, int G,
                                                              const double* __restrict__ xr64,
                                                              const float* __restrict__ coef,
                                                              const float* __restrict__ mean,
                                                              const float* __restrict__ cw, const float* __restrict__ cb,
                                                              float* __restrict__ dst, int accumulate) {
  __shared__ double sm[256];
  constexpr int C = B0_C, N = 9 * C;
  const int tid = threadIdx.x, cl = tid & 3, g = tid >> 2;
  const int n = blockIdx.x * 4 + cl;   // n = t * C + c
  double s = 0.0;
  if (n < N)
    for (int r = g; r < G; r += 64) s += (double)part_gx[(size_t)r * N + n];
  sm[tid] = s;
  __syncthreads();
  if (g == 0 && n < N) {
    double gx = 0.0;
    for (int k = 0; k < 64; ++k) gx += sm[k * 4 + cl];
    const int t = n / C, c = n % C;
    auto R = [&](int a, int b) {   // packed upper triangle behind the 9 entries of Sx
      const int lo = a < b ? a : b, hi = a < b ? b : a;
      return xr64[9 + lo * 9 - lo * (lo - 1) / 2 + (hi - lo)];
    };
    double yx = (double)cb[c] * xr64[t];
    for (int u = 0; u < 9; ++u) yx += (double)cw[c * 9 + u] * R(u, t);
    const double A = coef[c], Bc = coef[C + c], Cc = coef[2 * C + c];
    const double v = A * gx + Bc * (yx - (double)mean[c] * xr64[t]) + Cc * xr64[t];
    float* d = dst + c * 9 + t;
    *d = accumulate ? *d + (float)v : (float)v;
  }
}

// ---------------------------------------------------------------------------------------------
// C ABI
// ---------------------------------------------------------------------------------------------
extern "C" int bsed_block0_stats(const float* x, const float* cw, const float* cb, float* stats, float* xr_part,
                                 double* xr64, int G, int NB, int H, int W, int CO, void* stream) {
  BSED_CHECK_ARG(x && cw && cb && stats && xr_part && xr64, "bsed_block0_stats: null tensor");
  BSED_CHECK_ARG(CO == B0_C, "bsed_block0_stats: built for 16 first-layer channels (got %d)", CO);
  BSED_CHECK_ARG(G > 0 && NB > 0 && H > 0 && W > 0 && W <= 256 && (long)NB * H < (1L << 30),
                 "bsed_block0_stats: bad shape (W <= 256)");
  hipStream_t s = (hipStream_t)stream;
  hipLaunchKernelGGL(b0_stats_kernel, dim3(G), dim3(B0S_THREADS), 0, s, x, cw, cb, stats, xr_part, NB, H, W);
  hipLaunchKernelGGL(b0_colsum64_kernel, dim3(ceil_div(B0_NXR, 4)), dim3(256), 0, s, xr_part, G, B0_NXR, xr64);
  BSED_LAUNCH_CHECK();
  return BSED_OK;
}

extern "C" int bsed_block0_fwd(const float* x, const float* cw, const float* cb, const float* scale, const float* shift,
                               const float* wg, const float* bg, float* out, int B, int H, int W, int CO, int ph, int pw,
                               float drop_p, uint32_t rng_stream, uint64_t seed, void* stream) {
  BSED_CHECK_ARG(x && cw && cb && scale && shift && wg && bg && out, "bsed_block0_fwd: null tensor");
  BSED_CHECK_ARG(CO == B0_C, "bsed_block0_fwd: built for 16 first-layer channels (got %d)", CO);
  BSED_CHECK_ARG(B > 0 && H > 0 && W > 0 && (ph == 1 || ph == 2) && (pw == 1 || pw == 2) && W % pw == 0 && H >= ph,
                 "bsed_block0_fwd: bad shape");
  const long items = (long)B * (H / ph) * ((W + B0F_THREADS / 4 - 1) / (B0F_THREADS / 4));
  const dim3 grid((unsigned)std::min<long>(items, 8192));
  hipStream_t s = (hipStream_t)stream;
  if (ph == 2)
    hipLaunchKernelGGL(b0_fwd_kernel<2>, grid, dim3(B0F_THREADS), 0, s, x, cw, cb, scale, shift, wg, bg, out, B, H, W, pw,
                       drop_p, rng_stream, seed);
  else
    hipLaunchKernelGGL(b0_fwd_kernel<1>, grid, dim3(B0F_THREADS), 0, s, x, cw, cb, scale, shift, wg, bg, out, B, H, W, pw,
                       drop_p, rng_stream, seed);
  BSED_LAUNCH_CHECK();
  return BSED_OK;
}

extern "C" int bsed_block0_bwd(const float* x, const float* cw, const float* cb, const float* scale, const float* shift,
                               const float* wg, const float* bg, const float* dpool, float* part_dw, float* part_db,
                               float* part_st, float* part_gx, int G, int B, int H, int W, int CO, int ph, int pw,
                               float drop_p, uint32_t rng_stream, uint64_t seed, void* stream) {
  BSED_CHECK_ARG(x && cw && cb && scale && shift && wg && bg && dpool && part_dw && part_db && part_st && part_gx,
                 "bsed_block0_bwd: null tensor");
  BSED_CHECK_ARG(CO == B0_C, "bsed_block0_bwd: built for 16 first-layer channels (got %d)", CO);
  BSED_CHECK_ARG(B > 0 && H > 0 && W > 0 && G > 0 && (ph == 1 || ph == 2) && (pw == 1 || pw == 2) && W % pw == 0 &&
                     H >= ph, "bsed_block0_bwd: bad shape");
  BSED_CHECK_ARG((long)B * (H / ph) * ((W + 63) / 64) + G < (1L << 31), "bsed_block0_bwd: too many rows");
  hipStream_t s = (hipStream_t)stream;
  if (ph == 2)
    hipLaunchKernelGGL(b0_bwd_kernel<2>, dim3(G), dim3(B0B_THREADS), 0, s, x, cw, cb, scale, shift, wg, bg, dpool,
                       part_dw, part_db, part_st, part_gx, B, H, W, pw, drop_p, rng_stream, seed);
  else
    hipLaunchKernelGGL(b0_bwd_kernel<1>, dim3(G), dim3(B0B_THREADS), 0, s, x, cw, cb, scale, shift, wg, bg, dpool,
                       part_dw, part_db, part_st, part_gx, B, H, W, pw, drop_p, rng_stream, seed);
  BSED_LAUNCH_CHECK();
  return BSED_OK;
}

extern "C" int bsed_block0_wgrad_finish(const float* part_gx, int G, const double* xr64, const float* coef,
                                        const float* mean, const float* cw, const float* cb, float* dst, int accumulate,
                                        int CO, void* stream) {
  BSED_CHECK_ARG(part_gx && xr64 && coef && mean && cw && cb && dst && G > 0, "bsed_block0_wgrad_finish: bad argument");
  BSED_CHECK_ARG(CO == B0_C, "bsed_block0_wgrad_finish: built for 16 first-layer channels (got %d)", CO);
  hipLaunchKernelGGL(b0_wgrad_finish_kernel, dim3(9 * B0_C / 4), dim3(256), 0, (hipStream_t)stream, part_gx, G, xr64,
                     coef, mean, cw, cb, dst, accumulate);
  BSED_LAUNCH_CHECK();
  return BSED_OK;
}
