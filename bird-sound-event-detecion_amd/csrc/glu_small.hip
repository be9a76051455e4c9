// GLU stage of the FIRST CNN block (C = 16 channels on the full 865..1255 x 128 map) as pure streaming kernels.
//
// At C = 16 the per-position contraction is 16x16 = 256 MACs against 64 bytes of activation: 4 FLOP/B, far below
// the MFMA ridge, and the block owns the largest tensor of the network (1.8 GB at B = 256).  The MFMA tile
// kernel spends its time in per-tile overhead there, so this block gets dedicated HBM-bound kernels:
//   * 4 lanes per position, one float4 (4 channels) each -> perfectly coalesced 16 B/lane loads and stores;
//   * the 16-vector is all-gathered inside the 4-lane group with 3 xor-shuffles, the 16x16 weights sit in
//     registers pre-permuted to the shuffle order (no dynamic register indexing);
//   * forward:  y -> BN-apply -> Linear -> sigmoid gate -> dropout -> 2x2 avg-pool (vertical in registers,
//               horizontal by one shuffle) -> pooled                 [reference src/models/CNN.py:5-16,59-67]
//   * backward: y, d_pooled -> g = dL/d(BN output) written once, plus per-workgroup partials of dW_glu, db_glu and
//               the two BatchNorm-backward sums (sum g, sum g*y) -- one read of y, one write of g.
#include "bsed_common.h"
#include "../../include/bsed.h"
#include <algorithm>

#define GS_THREADS 512
#define GS_C 16

struct f4 { float v[4]; };

__device__ __forceinline__ f4 shfl_xor4(const f4& a, int s) {
  f4 r;
#pragma unroll
  for (int i = 0; i < 4; ++i) r.v[i] = __shfl_xor(a.v[i], s, 64);
  return r;
}

__global__ __launch_bounds__(GS_THREADS) void glu16_fwd_kernel(
    const float* __restrict__ y, const float* __restrict__ scale, const float* __restrict__ shift,
    const float* __restrict__ wg, const float* __restrict__ bg, float* __restrict__ out, int B, int H, int W, int ph,
    int pw, float drop_p, uint32_t rng_stream, uint64_t seed) {
  constexpr int C = GS_C;
  const int tid = threadIdx.x, q = tid & 3, col = tid >> 2;
  const int Hp = H / ph, Wp = W / pw;
  float wp[4][4][4];  // [shuffle step s][own channel i][kk] = W[4q+i][4(q^s)+kk]
#pragma unroll
  for (int s = 0; s < 4; ++s)
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
      for (int kk = 0; kk < 4; ++kk) wp[s][i][kk] = wg[(4 * q + i) * C + 4 * (q ^ s) + kk];
  float sc[4], sh[4], bi[4];
#pragma unroll
  for (int i = 0; i < 4; ++i) { sc[i] = scale[4 * q + i]; sh[i] = shift[4 * q + i]; bi[i] = bg[4 * q + i]; }
  const uint32_t dkey = drop_key(rng_stream, seed), dthr = drop_threshold(drop_p);
  const float dscale = drop_p > 0.f ? 1.0f / (1.0f - drop_p) : 1.0f;
  const float inv = 1.0f / (float)(ph * pw);

  for (long item = blockIdx.x; item < (long)B * Hp; item += gridDim.x) {
    const int b = (int)(item / Hp), hp = (int)(item % Hp);
    for (int w0 = 0; w0 < W; w0 += GS_THREADS / 4) {
      const int w = w0 + col;
      const bool ok = w < W;
      float pooled[4] = {0.f, 0.f, 0.f, 0.f};
      for (int dh = 0; dh < ph; ++dh) {
        const int h = hp * ph + dh;
        const size_t pos = ((size_t)b * H + h) * W + w;
        f4 xn;
        if (ok) {
          const float4 v = *reinterpret_cast<const float4*>(y + pos * C + 4 * q);
          xn.v[0] = fmaf(v.x, sc[0], sh[0]); xn.v[1] = fmaf(v.y, sc[1], sh[1]);
          xn.v[2] = fmaf(v.z, sc[2], sh[2]); xn.v[3] = fmaf(v.w, sc[3], sh[3]);
        } else {
          xn.v[0] = xn.v[1] = xn.v[2] = xn.v[3] = 0.f;
        }
        float lin[4] = {bi[0], bi[1], bi[2], bi[3]};
#pragma unroll
        for (int s = 0; s < 4; ++s) {
          const f4 o = s == 0 ? xn : shfl_xor4(xn, s);
#pragma unroll
          for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int kk = 0; kk < 4; ++kk) lin[i] = fmaf(wp[s][i][kk], o.v[kk], lin[i]);
        }
#pragma unroll
        for (int i = 0; i < 4; ++i)
          pooled[i] += lin[i] * sigmoid_fast(xn.v[i]) * drop_mul((uint64_t)pos * C + 4 * q + i, dkey, dthr, dscale);
      }
      if (pw == 2) {
#pragma unroll
        for (int i = 0; i < 4; ++i) pooled[i] += __shfl_xor(pooled[i], 4, 64);
      }
      if (ok && (w & (pw - 1)) == 0 && (w / pw) < Wp) {
        float4 o = make_float4(pooled[0] * inv, pooled[1] * inv, pooled[2] * inv, pooled[3] * inv);
        *reinterpret_cast<float4*>(out + (((size_t)b * Hp + hp) * Wp + w / pw) * C + 4 * q) = o;
      }
    }
  }
}

// MODE 0: g, db and the BN-backward sums (weights in registers, no dW accumulators)
// MODE 1: dW_glu only (64 accumulators, no weights) -- two passes keep both under 128 VGPRs with 4 waves/SIMD
// resident instead of one 256-VGPR kernel that spills and leaves HBM latency exposed.
template <int MODE>
__global__ __launch_bounds__(GS_THREADS) void glu16_bwd_kernel(
    const float* __restrict__ y, const float* __restrict__ scale, const float* __restrict__ shift,
    const float* __restrict__ wg, const float* __restrict__ bg, const float* __restrict__ dpool,
    float* __restrict__ g_out, float* __restrict__ part_dw /*(G,C,C)*/, float* __restrict__ part_db /*(G,2,C)*/,
    float* __restrict__ part_st /*(G,2,C)*/, int B, int H, int W, int ph, int pw, float drop_p, uint32_t rng_stream,
    uint64_t seed) {
  constexpr int C = GS_C;
  __shared__ float red[GS_THREADS * 17];
  const int tid = threadIdx.x, q = tid & 3, col = tid >> 2;
  const int Hp = H / ph, Wp = W / pw;
  const int sph = ph >> 1, spw = pw >> 1;
  // wp[s][i][kk] = W[4q+i][4(q^s)+kk].  It serves BOTH contractions: lin of the own channels (rows of W against the
  // all-gathered x) and g = d_lin W (own d_lin rows against all 16 columns, then a reduce-scatter inside the 4-lane
  // group), so the transposed copy of W never has to live in registers.
  float wp[4][4][4];
  if (MODE == 0) {
#pragma unroll
    for (int s = 0; s < 4; ++s)
#pragma unroll
      for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int kk = 0; kk < 4; ++kk) wp[s][i][kk] = wg[(4 * q + i) * C + 4 * (q ^ s) + kk];
  }
  float sc[4], sh[4], bi[4];
#pragma unroll
  for (int i = 0; i < 4; ++i) { sc[i] = scale[4 * q + i]; sh[i] = shift[4 * q + i]; bi[i] = bg[4 * q + i]; }
  const uint32_t dkey = drop_key(rng_stream, seed), dthr = drop_threshold(drop_p);
  const float dscale = drop_p > 0.f ? 1.0f / (1.0f - drop_p) : 1.0f;
  const float inv = 1.0f / (float)(ph * pw);

  float dwa[4][4][4];  // dW[4q+i][4(q^s)+kk]
  float dba[4] = {0.f, 0.f, 0.f, 0.f}, sga[4] = {0.f, 0.f, 0.f, 0.f}, sgya[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
  for (int s = 0; s < 4; ++s)
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
      for (int kk = 0; kk < 4; ++kk) dwa[s][i][kk] = 0.f;

  // flattened work list: (row of one clip, 128-column chunk); the next item's loads are issued before the
  // current item's arithmetic so HBM latency overlaps the shuffle/FMA chain
  const int chunks = (W + GS_THREADS / 4 - 1) / (GS_THREADS / 4);
  const int nitems = B * H * chunks;  // < 2^31 (checked on the host)
  float4 ny = make_float4(0.f, 0.f, 0.f, 0.f), nd = make_float4(0.f, 0.f, 0.f, 0.f);
  auto fetch = [&](int item) {
    ny = make_float4(0.f, 0.f, 0.f, 0.f);
    nd = make_float4(0.f, 0.f, 0.f, 0.f);
    if (item < nitems) {
      const int ch = item % chunks;
      const int row = item / chunks;
      const int h = row % H;
      const int b = row / H;
      const int w = ch * (GS_THREADS / 4) + col;
      if (w < W) {
        const size_t pos = ((size_t)b * H + h) * W + w;
        ny = *reinterpret_cast<const float4*>(y + pos * C + 4 * q);
        const int hp = h >> sph, wpi = w >> spw;
        if (hp < Hp && wpi < Wp)
          nd = *reinterpret_cast<const float4*>(dpool + (((size_t)b * Hp + hp) * Wp + wpi) * C + 4 * q);
      }
    }
  };
  fetch(blockIdx.x);
  for (int item = blockIdx.x; item < nitems; item += gridDim.x) {
    const float4 cy = ny, cd = nd;
    fetch(item + (int)gridDim.x);
    const int ch = item % chunks;
    const int row = item / chunks;
    const int h = row % H;
    const int b = row / H;
    const int w = ch * (GS_THREADS / 4) + col;
    const bool ok = w < W;
    const size_t pos = ((size_t)b * H + h) * W + w;
    const float yv[4] = {cy.x, cy.y, cy.z, cy.w};
    const float dres[4] = {cd.x, cd.y, cd.z, cd.w};
    f4 xs[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) xs[0].v[i] = ok ? fmaf(yv[i], sc[i], sh[i]) : 0.f;
#pragma unroll
    for (int s = 1; s < 4; ++s) xs[s] = shfl_xor4(xs[0], s);
    float lin[4] = {bi[0], bi[1], bi[2], bi[3]};
    if (MODE == 0) {
#pragma unroll
      for (int s = 0; s < 4; ++s)
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
          for (int kk = 0; kk < 4; ++kk) lin[i] = fmaf(wp[s][i][kk], xs[s].v[kk], lin[i]);
    }
    float dl[4], g[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const float sg = sigmoid_fast(xs[0].v[i]);
      const float dr = dres[i] * inv * drop_mul((uint64_t)pos * C + 4 * q + i, dkey, dthr, dscale);
      dl[i] = dr * sg;
      g[i] = dr * lin[i] * sg * (1.0f - sg);
    }
    // g[k] += sum_c d_lin[c] W[c][k]: own rows c against every column block, reduce-scatter over the group
#pragma unroll
    for (int s = 0; s < (MODE == 0 ? 4 : 0); ++s) {
      f4 ps;
#pragma unroll
      for (int kk = 0; kk < 4; ++kk) {
        float a = 0.f;
#pragma unroll
        for (int i = 0; i < 4; ++i) a = fmaf(dl[i], wp[s][i][kk], a);
        ps.v[kk] = a;
      }
      if (s > 0) ps = shfl_xor4(ps, s);
#pragma unroll
      for (int kk = 0; kk < 4; ++kk) g[kk] += ps.v[kk];
    }
    if (ok) {
      if (MODE == 0) {
        *reinterpret_cast<float4*>(g_out + pos * C + 4 * q) = make_float4(g[0], g[1], g[2], g[3]);
#pragma unroll
        for (int i = 0; i < 4; ++i) {
          dba[i] += dl[i];
          sga[i] += g[i];
          sgya[i] = fmaf(g[i], yv[i], sgya[i]);
        }
      } else {
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
          for (int s = 0; s < 4; ++s)
#pragma unroll
            for (int kk = 0; kk < 4; ++kk) dwa[s][i][kk] = fmaf(dl[i], xs[s].v[kk], dwa[s][i][kk]);
      }
    }
  }

  // workgroup reduction over the 128 threads that share a quad index q, 16 values per pass
#pragma unroll
  for (int pass = (MODE == 0 ? 4 : 0); pass < (MODE == 0 ? 5 : 4); ++pass) {
    __syncthreads();
#pragma unroll
    for (int j = 0; j < 16; ++j) {
      float v;
      if (pass < 4) v = dwa[pass][j >> 2][j & 3];
      else v = j < 4 ? dba[j & 3] : (j < 8 ? sga[j & 3] : (j < 12 ? sgya[j & 3] : 0.f));
      red[tid * 17 + j] = v;
    }
    __syncthreads();
    if (tid < 64) {
      const int rq = tid & 3, j = tid >> 2;
      float s = 0.f;
      for (int t = rq; t < GS_THREADS; t += 4) s += red[t * 17 + j];
      const size_t gblk = blockIdx.x;
      if (pass < 4) {
        part_dw[(gblk * C + 4 * rq + (j >> 2)) * C + 4 * (rq ^ pass) + (j & 3)] = s;
      } else if (j < 4) {
        part_db[(gblk * 2 + 0) * C + 4 * rq + j] = s;
        part_db[(gblk * 2 + 1) * C + 4 * rq + j] = 0.f;
      } else if (j < 8) {
        part_st[(gblk * 2 + 0) * C + 4 * rq + (j - 4)] = s;
      } else if (j < 12) {
        part_st[(gblk * 2 + 1) * C + 4 * rq + (j - 8)] = s;
      }
    }
  }
}

extern "C" int bsed_glu16_fwd(const float* y, const float* scale, const float* shift, const float* wg, const float* bg,
                              float* out, int B, int H, int W, int C, int ph, int pw, float drop_p, uint32_t rng_stream,
                              uint64_t seed, void* stream) {
  BSED_CHECK_ARG(y && scale && shift && wg && bg && out, "bsed_glu16_fwd: null tensor");
  BSED_CHECK_ARG(C == GS_C, "bsed_glu16_fwd: built for C=16 (got %d)", C);
  BSED_CHECK_ARG(B > 0 && H > 0 && W > 0 && (ph == 1 || ph == 2) && (pw == 1 || pw == 2) && W % pw == 0,
                 "bsed_glu16_fwd: bad shape");
  const long items = (long)B * (H / ph);
  BSED_CHECK_ARG(items > 0, "bsed_glu16_fwd: empty pooled extent");
  hipLaunchKernelGGL(glu16_fwd_kernel, dim3((unsigned)std::min<long>(items, 8192)), dim3(GS_THREADS), 0,
                     (hipStream_t)stream, y, scale, shift, wg, bg, out, B, H, W, ph, pw, drop_p, rng_stream, seed);
  BSED_LAUNCH_CHECK();
  return BSED_OK;
}

extern "C" int bsed_glu16_bwd(const float* y, const float* scale, const float* shift, const float* wg, const float* bg,
                              const float* dpool, float* g_out, float* part_dw, float* part_db, float* part_st, int G,
                              int B, int H, int W, int C, int ph, int pw, float drop_p, uint32_t rng_stream,
                              uint64_t seed, void* stream) {
  BSED_CHECK_ARG(y && scale && shift && wg && bg && dpool && g_out && part_dw && part_db && part_st,
                 "bsed_glu16_bwd: null tensor");
  BSED_CHECK_ARG(C == GS_C, "bsed_glu16_bwd: built for C=16 (got %d)", C);
  BSED_CHECK_ARG(B > 0 && H > 0 && W > 0 && G > 0 && (ph == 1 || ph == 2) && (pw == 1 || pw == 2) && W % pw == 0,
                 "bsed_glu16_bwd: bad shape");
  BSED_CHECK_ARG((long)B * H * ((W + 127) / 128) + G < (1L << 31), "bsed_glu16_bwd: too many rows");
  hipLaunchKernelGGL(glu16_bwd_kernel<0>, dim3(G), dim3(GS_THREADS), 0, (hipStream_t)stream, y, scale, shift, wg, bg,
                     dpool, g_out, part_dw, part_db, part_st, B, H, W, ph, pw, drop_p, rng_stream, seed);
  hipLaunchKernelGGL(glu16_bwd_kernel<1>, dim3(G), dim3(GS_THREADS), 0, (hipStream_t)stream, y, scale, shift, wg, bg,
                     dpool, g_out, part_dw, part_db, part_st, B, H, W, ph, pw, drop_p, rng_stream, seed);
  BSED_LAUNCH_CHECK();
  return BSED_OK;
}
