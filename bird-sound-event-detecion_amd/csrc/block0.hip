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
#include <stdlib.h>

#define B0_C 16
#define B0_NXR 54   // Sx[9] then R packed upper triangle (t <= t'): 45

// THE definition of y0: bias, then taps 0..8 in order, one fma each (conv0_fwd_kernel uses the same chain)
__device__ __forceinline__ float b0_conv1(const float (&x9)[9], const float (&w)[9], float b) {
  float a = b;
#pragma unroll
  for (int t = 0; t < 9; ++t) a = fmaf(x9[t], w[t], a);
  return a;
}

__device__ __forceinline__ f32x4 b0_mm16(const float (&a)[4], float b0, float b1, float b2, float b3, f32x4 c) {
  asm volatile("s_nop 4" : "+v"(b0), "+v"(b1), "+v"(b2), "+v"(b3), "+v"(c));
  c = __builtin_amdgcn_mfma_f32_16x16x4f32(a[0], b0, c, 0, 0, 0);
  c = __builtin_amdgcn_mfma_f32_16x16x4f32(a[1], b1, c, 0, 0, 0);
  c = __builtin_amdgcn_mfma_f32_16x16x4f32(a[2], b2, c, 0, 0, 0);
  c = __builtin_amdgcn_mfma_f32_16x16x4f32(a[3], b3, c, 0, 0, 0);
  return c;
}

// bf16 mode (ABF instances): the same 16 x 16 x 16 contraction as ONE v_mfma_f32_16x16x16_bf16 -- operands rounded to bf16
// (round to nearest even), fp32 accumulation; ~19 cycles instead of 4 x 32 (tools/probe/mfma16_probe.hip).  The lane layout
// is the fp32 chain's: a lane's four values are k = 4 q .. 4 q + 3 of row / column p.
typedef short b0_s16x4 __attribute__((ext_vector_type(4)));
__device__ __forceinline__ b0_s16x4 b0_pack4(float v0, float v1, float v2, float v3) {
  const f32x2 a = {v0, v1}, b = {v2, v3};
  const uint2 r = make_uint2(__builtin_bit_cast(uint32_t, __builtin_convertvector(a, bsed_bf16x2)),
                             __builtin_bit_cast(uint32_t, __builtin_convertvector(b, bsed_bf16x2)));
  return __builtin_bit_cast(b0_s16x4, r);
}
__device__ __forceinline__ f32x4 b0_mm16_bf(b0_s16x4 a, float b0, float b1, float b2, float b3, f32x4 c) {
  b0_s16x4 b = b0_pack4(b0, b1, b2, b3);
  asm volatile("s_nop 4" : "+v"(b), "+v"(c));
  return __builtin_amdgcn_mfma_f32_16x16x16bf16_1k(a, b, c, 0, 0, 0);
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
// cwt = the conv weight TRANSPOSED, [tap][channel]: the channels of one tap are then consecutive scalar registers and
// a pair of them is one operand of v_pk_fma_f32 (two channels per instruction; with the (16,1,3,3) layout the compiler
// packed over channels all the same and paid ~200 v_mov per position to build the pairs: 0.26 -> 0.13 ms)
__global__ __launch_bounds__(B0S_THREADS) void b0_stats_kernel(const float* __restrict__ x, const float* __restrict__ cwt,
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
      for (int t = 0; t < 9; ++t) wv[t] = cwt[t * B0_C + c];   // wave-uniform: scalar loads, SGPR-pair operands
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
// Shared pieces of the forward / backward kernels.
//
// A wave owns 16 consecutive columns of PH stacked rows.  The (PH + 2) x 18 input window around them (zero outside the
// map) lives in a WAVE-PRIVATE LDS tile: two global loads per lane and item (issued one item ahead), no workgroup
// barrier -- LDS operations of one wave execute in order.  The convolution runs on the fp32 matrix cores like the GLU
// contractions:  y^T[c][pos] = sum_t W[c][t] x_t[pos]  as three v_mfma_f32_16x16x4_f32 (taps 0-3, 4-7, 8 + padding)
// with A[i = c][k] = W[c = p][4r + q] (lane constants), B[k][j = pos] = x_{4r+q}[pos p] (ONE LDS read per lane and
// MFMA: the lane's tap offset is a lane constant) and the bias as initial accumulator; the D fragment is the lane's own
// position and four channels.  Same accumulation chain in every kernel of this file.
// ---------------------------------------------------------------------------------------------
#define B0_WIN 18   // window columns: 16 + halo

template <int PH>
struct B0Win {
  static constexpr int NE = (PH + 2) * B0_WIN;   // 72 (PH = 2) or 54 floats
  static constexpr int NL = (NE + 63) / 64;      // loads per lane and item
  int ej[NL], ec[NL];                            // (row, column) of the lane's window elements
  bool ev[NL];
  float v[NL];                                   // prefetched values
  __device__ __forceinline__ void init(int lane) {
#pragma unroll
    for (int k = 0; k < NL; ++k) {
      const int e = lane + 64 * k;
      ev[k] = e < NE;
      ej[k] = (e < NE ? e : 0) / B0_WIN;
      ec[k] = (e < NE ? e : 0) % B0_WIN;
    }
  }
  // issue the loads of the window whose first row / column are (r0 - 1, c0 - 1) in image b
  // (NB * H * W < 2^31, checked on the host; SMALL: the tensor is below 4 GB, so the address is the uniform base plus
  //  a 32-bit BYTE offset -- one VALU instruction instead of a 64-bit multiply-add chain per load)
  template <bool SMALL>
  __device__ __forceinline__ void fetch(const float* __restrict__ x, int b, int r0, int c0, int H, int W) {
    const int base = b * H;
#pragma unroll
    for (int k = 0; k < NL; ++k) {
      const int r = r0 - 1 + ej[k], c = c0 - 1 + ec[k];
      const bool ok = r >= 0 && r < H && c >= 0 && c < W;
      const int idx = (base + min(max(r, 0), H - 1)) * W + min(max(c, 0), W - 1);
      float t;
      if (SMALL) t = *reinterpret_cast<const float*>(reinterpret_cast<const char*>(x) + ((uint32_t)idx << 2));
      else t = x[idx];
      v[k] = ok ? t : 0.f;
    }
  }
  __device__ __forceinline__ void store(float* win, int lane) const {
#pragma unroll
    for (int k = 0; k < NL; ++k)
      if (ev[k]) win[lane + 64 * k] = v[k];
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");   // the wave's own reads below see the whole tile
  }
};

// conv weights as MFMA A operands: aw[r] = W[c = p][t = 4r + q] (0 for t >= 9); window offsets of the lane's taps
struct B0Conv {
  float aw[3];
  int off[3];      // (kh * 18 + kw) of tap min(4r + q, 8), plus the lane's column p
  float bias[4];
  __device__ __forceinline__ void init(const float* __restrict__ cw, const float* __restrict__ cb, int p, int q) {
#pragma unroll
    for (int r = 0; r < 3; ++r) {
      const int t = 4 * r + q, tc = t < 9 ? t : 8;
      aw[r] = t < 9 ? cw[p * 9 + t] : 0.f;
      off[r] = (tc / 3) * B0_WIN + (tc % 3) + p;
    }
#pragma unroll
    for (int i = 0; i < 4; ++i) bias[i] = cb[4 * q + i];
  }
  // bf16 mode: taps 4q .. 4q+3 as one packed operand (taps >= 9 carry weight 0 and read tap 8's window element)
  b0_s16x4 aw16;
  int off16[4];
  __device__ __forceinline__ void init_bf(const float* __restrict__ cw, int p, int q) {
    float w4[4];
#pragma unroll
    for (int kk = 0; kk < 4; ++kk) {
      const int t = 4 * q + kk, tc = t < 9 ? t : 8;
      w4[kk] = t < 9 ? cw[p * 9 + t] : 0.f;
      off16[kk] = (tc / 3) * B0_WIN + (tc % 3) + p;
    }
    aw16 = b0_pack4(w4[0], w4[1], w4[2], w4[3]);
  }
  __device__ __forceinline__ f32x4 run_bf(const float* win, int dh) const {
    const float b0 = win[dh * B0_WIN + off16[0]], b1 = win[dh * B0_WIN + off16[1]], b2 = win[dh * B0_WIN + off16[2]],
                b3 = win[dh * B0_WIN + off16[3]];
    return b0_mm16_bf(aw16, b0, b1, b2, b3, f32x4{bias[0], bias[1], bias[2], bias[3]});
  }
  // y of the lane's position in window row dh (its four channels 4q .. 4q+3)
  __device__ __forceinline__ f32x4 run(const float* win, int dh) const {
    float b0 = win[dh * B0_WIN + off[0]], b1 = win[dh * B0_WIN + off[1]], b2 = win[dh * B0_WIN + off[2]];
    f32x4 c = {bias[0], bias[1], bias[2], bias[3]};
    asm volatile("s_nop 4" : "+v"(b0), "+v"(b1), "+v"(b2), "+v"(c));
    c = __builtin_amdgcn_mfma_f32_16x16x4f32(aw[0], b0, c, 0, 0, 0);
    c = __builtin_amdgcn_mfma_f32_16x16x4f32(aw[1], b1, c, 0, 0, 0);
    c = __builtin_amdgcn_mfma_f32_16x16x4f32(aw[2], b2, c, 0, 0, 0);
    return c;
  }
};

// ---------------------------------------------------------------------------------------------
// forward: x -> conv -> BN-apply -> Linear -> gate -> dropout -> PH x pw average pool -> pooled
// ---------------------------------------------------------------------------------------------
#define B0F_THREADS 512
template <int PH, bool SMALL, int ABF>   // ABF: the pooled output is a bf16 tensor (bf16 mode)
__global__ __launch_bounds__(B0F_THREADS) void b0_fwd_kernel(
    const float* __restrict__ x, const float* __restrict__ cw, const float* __restrict__ cb,
    const float* __restrict__ scale, const float* __restrict__ shift, const float* __restrict__ wg,
    const float* __restrict__ bg, float* __restrict__ out, int B, int H, int W, int pw, float drop_p,
    uint32_t rng_stream, uint64_t seed, const uint64_t* __restrict__ seed_add) {
  if (seed_add) seed += *seed_add;
  constexpr int C = B0_C;
  __shared__ float wins[B0F_THREADS / 64][(PH + 2) * B0_WIN + 8];
  const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63, p = lane & 15, q = lane >> 4;
  const int col = wave * 16 + p;
  const int Hp = H / PH, Wp = W / pw;
  float* win = wins[wave];
  B0Conv K;
  K.init(cw, cb, p, q);
  if (ABF) K.init_bf(cw, p, q);
  float a1[4];
#pragma unroll
  for (int kk = 0; kk < 4; ++kk) a1[kk] = wg[p * C + 4 * q + kk];
  const b0_s16x4 a1b = b0_pack4(a1[0], a1[1], a1[2], a1[3]);
  float sc[4], sh[4], bi[4];
#pragma unroll
  for (int i = 0; i < 4; ++i) { sc[i] = scale[4 * q + i]; sh[i] = shift[4 * q + i]; bi[i] = bg[4 * q + i]; }
  const uint32_t dkey = drop_key(rng_stream, seed), dthr = drop_threshold16(drop_p);
  const float dscale = drop_p > 0.f ? 1.0f / (1.0f - drop_p) : 1.0f;
  const float inv = 1.0f / (float)(PH * pw);
  const int chunks = (W + B0F_THREADS / 4 - 1) / (B0F_THREADS / 4);
  B0Idx cur, nxt;
  cur.init(blockIdx.x, gridDim.x, Hp, chunks);
  B0Win<PH> X;
  X.init(lane);
  X.template fetch<SMALL>(x, cur.b < B ? cur.b : 0, cur.hp * PH, cur.ch * (B0F_THREADS / 4) + wave * 16, H, W);
  for (; cur.b < B; cur = nxt) {
    X.store(win, lane);
    nxt = cur.next();
    {
      const B0Idx& I = nxt.b < B ? nxt : cur;   // the last prefetch re-reads the current item: no branch
      X.template fetch<SMALL>(x, I.b, I.hp * PH, I.ch * (B0F_THREADS / 4) + wave * 16, H, W);
    }
    const int hp = cur.hp, b = cur.b;
    const int w = cur.ch * (B0F_THREADS / 4) + col;
    const bool ok = w < W;
    const int wc = ok ? w : W - 1;
    float pooled[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int dh = 0; dh < PH; ++dh) {
      const int h = hp * PH + dh;
      const size_t pos = ((size_t)b * H + h) * W + wc;
      const uint32_t pos32 = (uint32_t)((b * H + h) * W + wc);
      const f32x4 y = ABF ? K.run_bf(win, dh) : K.run(win, dh);
      float xn[4];
#pragma unroll
      for (int i = 0; i < 4; ++i) xn[i] = fmaf(y[i], sc[i], sh[i]);
      const f32x4 lin = ABF ? b0_mm16_bf(a1b, xn[0], xn[1], xn[2], xn[3], f32x4{0.f, 0.f, 0.f, 0.f})
                            : b0_mm16(a1, xn[0], xn[1], xn[2], xn[3], f32x4{0.f, 0.f, 0.f, 0.f});
      float dm[4];
      if (SMALL) {   // fewer than 2^32 elements: 32-bit counters (same masks)
        drop_mul2_32(pos32 * C + 4 * q, dkey, dthr, dscale, dm[0], dm[1]);
        drop_mul2_32(pos32 * C + 4 * q + 2, dkey, dthr, dscale, dm[2], dm[3]);
      } else {
        drop_mul2((uint64_t)pos * C + 4 * q, dkey, dthr, dscale, dm[0], dm[1]);
        drop_mul2((uint64_t)pos * C + 4 * q + 2, dkey, dthr, dscale, dm[2], dm[3]);
      }
#pragma unroll
      for (int i = 0; i < 4; ++i) pooled[i] += (lin[i] + bi[i]) * sigmoid_fast(xn[i]) * dm[i];
    }
    if (pw == 2) {
#pragma unroll
      for (int i = 0; i < 4; ++i) pooled[i] += __shfl_xor(pooled[i], 1, 64);   // the neighbouring column
    }
    if (ok && (w & (pw - 1)) == 0 && (w >> (pw >> 1)) < Wp) {   // pw is 1 or 2
      const float4 o = make_float4(pooled[0] * inv, pooled[1] * inv, pooled[2] * inv, pooled[3] * inv);
      if (ABF) {
        act_st4<ABF>(out, (((size_t)b * Hp + hp) * Wp + (w >> (pw >> 1))) * C + 4 * q, f32x4{o.x, o.y, o.z, o.w});
      } else if (SMALL) {
        const uint32_t ob = (uint32_t)((((b * Hp + hp) * Wp + (w >> (pw >> 1))) * C + 4 * q) << 2);
        *reinterpret_cast<float4*>(reinterpret_cast<char*>(out) + ob) = o;
      } else {
        *reinterpret_cast<float4*>(out + (((size_t)b * Hp + hp) * Wp + (w >> (pw >> 1))) * C + 4 * q) = o;
      }
    }
  }
}

// ---------------------------------------------------------------------------------------------
// backward: x, d_pooled -> partials of dW_glu (G,16,16), db_glu (G,2,16), (sum g, sum g y) (G,2,16), Gx (G,9,16)
//
// dW_glu[n][c] = sum_pos d_lin[pos][n] xn[pos][c] contracts over POSITIONS, which the lane map keeps in the lane index:
// both operands go through a wave-private LDS tile ([16 positions][16 channels], pitch 20: the float4 row stores and
// the transposed dword reads are both conflict-free) and come back with positions in the MFMA's k index (lane quarter
// q <-> position 4q + s for MFMA s), channel in p: four more v_mfma_f32_16x16x4_f32 per 16 positions, ONE four-register
// accumulator per lane instead of 64 VALU accumulators and 12 cross-lane shuffles per position.
// ---------------------------------------------------------------------------------------------
#define B0B_THREADS 256
#ifndef B0B_WPE
#define B0B_WPE 3
#endif
#define B0_TP 20   // pitch of the transpose tiles (floats)

template <int PH, bool SMALL, int ABF>   // ABF: d_pooled is a bf16 tensor (bf16 mode)
__global__ __launch_bounds__(B0B_THREADS, B0B_WPE) void b0_bwd_kernel(
    const float* __restrict__ x, const float* __restrict__ cw, const float* __restrict__ cb,
    const float* __restrict__ scale, const float* __restrict__ shift, const float* __restrict__ wg,
    const float* __restrict__ bg, const float* __restrict__ dpool, float* __restrict__ part_dw,
    float* __restrict__ part_db, float* __restrict__ part_st, float* __restrict__ part_gx, int B, int H, int W, int pw,
    float drop_p, uint32_t rng_stream, uint64_t seed, const uint64_t* __restrict__ seed_add) {
  if (seed_add) seed += *seed_add;
  constexpr int C = B0_C;
  __shared__ float red[B0B_THREADS * 17];
  __shared__ float wins[B0B_THREADS / 64][(PH + 2) * B0_WIN + 8];
  __shared__ __align__(16) float tiles[B0B_THREADS / 64][2][16 * B0_TP];
  const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63, p = lane & 15, q = lane >> 4;
  const int col = wave * 16 + p;
  const int Hp = H / PH, Wp = W / pw;
  const int spw = pw >> 1;
  float* win = wins[wave];
  float* tdl = tiles[wave][0];
  float* txn = tiles[wave][1];
  B0Conv K;
  K.init(cw, cb, p, q);
  if (ABF) K.init_bf(cw, p, q);
  int xoff[9];   // the nine taps at the lane's own position (VALU Gx)
#pragma unroll
  for (int t = 0; t < 9; ++t) xoff[t] = (t / 3) * B0_WIN + (t % 3) + p;
  float a1[4], a2[4];   // W[n = p][4q + kk]  and  W[4q + kk][c = p]
#pragma unroll
  for (int kk = 0; kk < 4; ++kk) { a1[kk] = wg[p * C + 4 * q + kk]; a2[kk] = wg[(4 * q + kk) * C + p]; }
  const b0_s16x4 a1b = b0_pack4(a1[0], a1[1], a1[2], a1[3]), a2b = b0_pack4(a2[0], a2[1], a2[2], a2[3]);
  float sc[4], sh[4], bi[4];
#pragma unroll
  for (int i = 0; i < 4; ++i) { sc[i] = scale[4 * q + i]; sh[i] = shift[4 * q + i]; bi[i] = bg[4 * q + i]; }
  const uint32_t dkey = drop_key(rng_stream, seed), dthr = drop_threshold16(drop_p);
  const float dscale = drop_p > 0.f ? 1.0f / (1.0f - drop_p) : 1.0f;
  const float inv = 1.0f / (float)(PH * pw);

  f32x4 dwacc = {0.f, 0.f, 0.f, 0.f};   // dW[n = 4q + v][c = p]
  float gxa[4][9];                      // Gx[4q+i][t]
  float dba[4] = {0.f, 0.f, 0.f, 0.f}, sga[4] = {0.f, 0.f, 0.f, 0.f}, sgya[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int t = 0; t < 9; ++t) gxa[i][t] = 0.f;

  // work list: (image, pooled row, 64-column chunk); rows beyond Hp * PH have zero gradient and are not visited
  const int chunks = (W + B0B_THREADS / 4 - 1) / (B0B_THREADS / 4);
  B0Idx cur, nxt;
  cur.init(blockIdx.x, gridDim.x, Hp, chunks);
  B0Win<PH> X;
  X.init(lane);
  float4 nd = make_float4(0.f, 0.f, 0.f, 0.f);
  auto fetch = [&](const B0Idx& I) {
    X.template fetch<SMALL>(x, I.b, I.hp * PH, I.ch * (B0B_THREADS / 4) + wave * 16, H, W);
    const int w = min(I.ch * (B0B_THREADS / 4) + col, W - 1);
    const int wpi = min(w >> spw, Wp - 1);
    if (ABF) {
      const f32x4 t = act_ld4<ABF>(dpool, (((size_t)I.b * Hp + I.hp) * Wp + wpi) * C + 4 * q);
      nd = make_float4(t[0], t[1], t[2], t[3]);
    } else if (SMALL) {
      const uint32_t ob = (uint32_t)((((I.b * Hp + I.hp) * Wp + wpi) * C + 4 * q) << 2);
      nd = *reinterpret_cast<const float4*>(reinterpret_cast<const char*>(dpool) + ob);
    } else {
      nd = *reinterpret_cast<const float4*>(dpool + (((size_t)I.b * Hp + I.hp) * Wp + wpi) * C + 4 * q);
    }
  };
  if (cur.b < B) fetch(cur);
  for (; cur.b < B; cur = nxt) {
    X.store(win, lane);
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
      const uint32_t pos32 = (uint32_t)((b * H + h) * W + min(w, W - 1));
      const f32x4 yv = ABF ? K.run_bf(win, dh) : K.run(win, dh);
      float xn[4];
#pragma unroll
      for (int i = 0; i < 4; ++i) xn[i] = fmaf(yv[i], sc[i], sh[i]);
      const f32x4 lin = ABF ? b0_mm16_bf(a1b, xn[0], xn[1], xn[2], xn[3], f32x4{0.f, 0.f, 0.f, 0.f})
                            : b0_mm16(a1, xn[0], xn[1], xn[2], xn[3], f32x4{0.f, 0.f, 0.f, 0.f});
      float dl[4], dm[4];
      f32x4 gt;
      if (SMALL) {
        drop_mul2_32(pos32 * C + 4 * q, dkey, dthr, dscale, dm[0], dm[1]);
        drop_mul2_32(pos32 * C + 4 * q + 2, dkey, dthr, dscale, dm[2], dm[3]);
      } else {
        drop_mul2((uint64_t)pos * C + 4 * q, dkey, dthr, dscale, dm[0], dm[1]);
        drop_mul2((uint64_t)pos * C + 4 * q + 2, dkey, dthr, dscale, dm[2], dm[3]);
      }
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const float sg = sigmoid_fast(xn[i]);
        const float dr = dres[i] * dm[i];
        dl[i] = dr * sg;
        gt[i] = dr * (lin[i] + bi[i]) * sg * (1.0f - sg);
      }
      // operands of the dW contraction into the transpose tiles (row = position p, columns 4q .. 4q+3)
      *reinterpret_cast<float4*>(tdl + p * B0_TP + 4 * q) = make_float4(dl[0], dl[1], dl[2], dl[3]);
      *reinterpret_cast<float4*>(txn + p * B0_TP + 4 * q) = make_float4(xn[0], xn[1], xn[2], xn[3]);
      // g = d_lin W + gate term (0 on idle columns)
      const f32x4 g = ABF ? b0_mm16_bf(a2b, dl[0], dl[1], dl[2], dl[3], gt) : b0_mm16(a2, dl[0], dl[1], dl[2], dl[3], gt);
      float x9[9];
#pragma unroll
      for (int t = 0; t < 9; ++t) x9[t] = win[dh * B0_WIN + xoff[t]];
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");   // tiles written (all lanes of the wave), taps read
      float ta[4], tb[4];
#pragma unroll
      for (int s = 0; s < 4; ++s) { ta[s] = tdl[(4 * q + s) * B0_TP + p]; tb[s] = txn[(4 * q + s) * B0_TP + p]; }
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        dba[i] += dl[i];
        sga[i] += g[i];
        sgya[i] = fmaf(g[i], yv[i], sgya[i]);
#pragma unroll
        for (int t = 0; t < 9; ++t) gxa[i][t] = fmaf(g[i], x9[t], gxa[i][t]);
      }
      asm volatile("s_waitcnt lgkmcnt(0)\n\ts_nop 4" : "+v"(ta[0]), "+v"(ta[1]), "+v"(ta[2]), "+v"(ta[3]), "+v"(tb[0]),
                   "+v"(tb[1]), "+v"(tb[2]), "+v"(tb[3]), "+v"(dwacc) : : "memory");
      if (ABF) {
        b0_s16x4 ta4 = b0_pack4(ta[0], ta[1], ta[2], ta[3]), tb4 = b0_pack4(tb[0], tb[1], tb[2], tb[3]);
        asm volatile("s_nop 4" : "+v"(ta4), "+v"(tb4), "+v"(dwacc));
        dwacc = __builtin_amdgcn_mfma_f32_16x16x16bf16_1k(ta4, tb4, dwacc, 0, 0, 0);
      } else {
#pragma unroll
        for (int s = 0; s < 4; ++s) dwacc = __builtin_amdgcn_mfma_f32_16x16x4f32(ta[s], tb[s], dwacc, 0, 0, 0);
      }
    }
  }

  // workgroup reduction.  pass 0: the waves' dW accumulators (already summed over positions); passes 1-4: values kept
  // per lane, summed over the 64 threads that share a quarter index q, 16 values per pass (db / sum g / sum g y, Gx)
  const size_t gblk = blockIdx.x;
  __syncthreads();
#pragma unroll
  for (int v = 0; v < 4; ++v) red[(wave * 4 + v) * 64 + lane] = dwacc[v];
  __syncthreads();
  {
    const int v = tid >> 6, l = tid & 63;   // element (n = 4 * (l >> 4) + v, c = l & 15)
    float s = 0.f;
    for (int wv = 0; wv < B0B_THREADS / 64; ++wv) s += red[(wv * 4 + v) * 64 + l];
    part_dw[(gblk * C + 4 * (l >> 4) + v) * C + (l & 15)] = s;
  }
#pragma unroll
  for (int pass = 0; pass < 4; ++pass) {
    __syncthreads();
#pragma unroll
    for (int j = 0; j < 16; ++j) {
      float v;
      if (pass == 0) v = j < 4 ? dba[j & 3] : (j < 8 ? sga[j & 3] : (j < 12 ? sgya[j & 3] : 0.f));
      else {
        const int e = (pass - 1) * 12 + j;   // e = i * 9 + t, 36 values over three passes of 12
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
      if (pass == 0) {
        if (j < 4) {
          part_db[(gblk * 2 + 0) * C + 4 * rq + j] = s;
          part_db[(gblk * 2 + 1) * C + 4 * rq + j] = 0.f;
        } else if (j < 8) {
          part_st[(gblk * 2 + 0) * C + 4 * rq + (j - 4)] = s;
        } else if (j < 12) {
          part_st[(gblk * 2 + 1) * C + 4 * rq + (j - 8)] = s;
        }
      } else if (j < 12) {
        const int e = (pass - 1) * 12 + j, i = e / 9, t = e % 9;
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
extern "C" int bsed_block0_stats(const float* x, const float* cw_t, const float* cb, float* stats, float* xr_part,
                                 double* xr64, int G, int NB, int H, int W, int CO, void* stream) {
  BSED_CHECK_ARG(x && cw_t && cb && stats && xr_part && xr64, "bsed_block0_stats: null tensor");
  BSED_CHECK_ARG(CO == B0_C, "bsed_block0_stats: built for 16 first-layer channels (got %d)", CO);
  BSED_CHECK_ARG(G > 0 && NB > 0 && H > 0 && W > 0 && W <= 256 && (long)NB * H < (1L << 30),
                 "bsed_block0_stats: bad shape (W <= 256)");
  hipStream_t s = (hipStream_t)stream;
  hipLaunchKernelGGL(b0_stats_kernel, dim3(G), dim3(B0S_THREADS), 0, s, x, cw_t, cb, stats, xr_part, NB, H, W);
  hipLaunchKernelGGL(b0_colsum64_kernel, dim3(ceil_div(B0_NXR, 4)), dim3(256), 0, s, xr_part, G, B0_NXR, xr64);
  BSED_LAUNCH_CHECK();
  return BSED_OK;
}

extern "C" int bsed_block0_fwd(const float* x, const float* cw, const float* cb, const float* scale, const float* shift,
                               const float* wg, const float* bg, float* out, int B, int H, int W, int CO, int ph, int pw,
                               float drop_p, uint32_t rng_stream, uint64_t seed, int act_bf16, void* stream) {
  BSED_CHECK_ARG(x && cw && cb && scale && shift && wg && bg && out, "bsed_block0_fwd: null tensor");
  BSED_CHECK_ARG(CO == B0_C, "bsed_block0_fwd: built for 16 first-layer channels (got %d)", CO);
  BSED_CHECK_ARG(B > 0 && H > 0 && W > 0 && (ph == 1 || ph == 2) && (pw == 1 || pw == 2) && W % pw == 0 && H >= ph &&
                     (long)B * H * W < (1L << 31), "bsed_block0_fwd: bad shape");
  const long items = (long)B * (H / ph) * ((W + B0F_THREADS / 4 - 1) / (B0F_THREADS / 4));
  static const long gmax = getenv("BSED_B0_FWD_G") ? atol(getenv("BSED_B0_FWD_G")) : 8192;   // A/B knob
  const dim3 grid((unsigned)std::min<long>(items, gmax));
  hipStream_t s = (hipStream_t)stream;
  // SMALL: fewer than 2^28 positions (every tensor below 4 GB, element counters below 2^32): 32-bit offsets
  const bool small = (long)B * H * W < (1L << 28);
#define B0_LAUNCH_FWD(PH_, SM_, AB_)                                                                                 \
  hipLaunchKernelGGL((b0_fwd_kernel<PH_, SM_, AB_>), grid, dim3(B0F_THREADS), 0, s, x, cw, cb, scale, shift, wg, bg, out, B, \
                     H, W, pw, drop_p, rng_stream, seed, bsed_seed_add_ptr())
  if (act_bf16) { if (ph == 2) B0_LAUNCH_FWD(2, false, 1); else B0_LAUNCH_FWD(1, false, 1); }
  else if (ph == 2) { if (small) B0_LAUNCH_FWD(2, true, 0); else B0_LAUNCH_FWD(2, false, 0); }
  else { if (small) B0_LAUNCH_FWD(1, true, 0); else B0_LAUNCH_FWD(1, false, 0); }
#undef B0_LAUNCH_FWD
  BSED_LAUNCH_CHECK();
  return BSED_OK;
}

extern "C" int bsed_block0_bwd(const float* x, const float* cw, const float* cb, const float* scale, const float* shift,
                               const float* wg, const float* bg, const float* dpool, float* part_dw, float* part_db,
                               float* part_st, float* part_gx, int G, int B, int H, int W, int CO, int ph, int pw,
                               float drop_p, uint32_t rng_stream, uint64_t seed, int act_bf16, void* stream) {
  BSED_CHECK_ARG(x && cw && cb && scale && shift && wg && bg && dpool && part_dw && part_db && part_st && part_gx,
                 "bsed_block0_bwd: null tensor");
  BSED_CHECK_ARG(CO == B0_C, "bsed_block0_bwd: built for 16 first-layer channels (got %d)", CO);
  BSED_CHECK_ARG(B > 0 && H > 0 && W > 0 && G > 0 && (ph == 1 || ph == 2) && (pw == 1 || pw == 2) && W % pw == 0 &&
                     H >= ph, "bsed_block0_bwd: bad shape");
  BSED_CHECK_ARG((long)B * H * W < (1L << 31), "bsed_block0_bwd: too many positions");
  hipStream_t s = (hipStream_t)stream;
  const bool small = (long)B * H * W < (1L << 28);
#define B0_LAUNCH_BWD(PH_, SM_, AB_)                                                                                  \
  hipLaunchKernelGGL((b0_bwd_kernel<PH_, SM_, AB_>), dim3(G), dim3(B0B_THREADS), 0, s, x, cw, cb, scale, shift, wg, bg, dpool, \
                     part_dw, part_db, part_st, part_gx, B, H, W, pw, drop_p, rng_stream, seed, bsed_seed_add_ptr())
  if (act_bf16) { if (ph == 2) { if (small) B0_LAUNCH_BWD(2, true, 1); else B0_LAUNCH_BWD(2, false, 1); }
                  else { if (small) B0_LAUNCH_BWD(1, true, 1); else B0_LAUNCH_BWD(1, false, 1); } }
  else if (ph == 2) { if (small) B0_LAUNCH_BWD(2, true, 0); else B0_LAUNCH_BWD(2, false, 0); }
  else { if (small) B0_LAUNCH_BWD(1, true, 0); else B0_LAUNCH_BWD(1, false, 0); }
#undef B0_LAUNCH_BWD
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
