// GLU stage of the FIRST CNN block (C = 16 channels on the full 865..1255 x 128 map), the largest tensor of the
// network (1.8 GB at B = 256), as streaming kernels: one float4 (4 channels) per lane, 16 B coalesced loads/stores.
//   * forward:  y -> BN-apply -> Linear -> sigmoid gate -> dropout -> 2x2 avg-pool (vertical in registers,
//               horizontal by one shuffle) -> pooled                 [reference src/models/CNN.py:5-16,59-67]
//   * backward: y, d_pooled -> g = dL/d(BN output) written once, plus per-workgroup partials of dW_glu, db_glu and
//               the two BatchNorm-backward sums (sum g, sum g*y) -- ONE pass: one read of y, one write of g.
// The per-position 16x16 contractions run on the fp32 matrix cores (v_mfma_f32_16x16x4_f32, exact fp32) thanks to the
// lane map described below; the first version did them as register matvecs with shuffles and was VALU-bound
// (0.68 + 2.23 ms per step at B = 256; now 0.48 + 0.97).
#include "bsed_common.h"
#include "../../include/bsed.h"
#include <algorithm>

#define GS_THREADS 512
#define GS_C 16

// Lane map of both kernels: lane = 16*q + p, p = one of the wave's 16 consecutive columns, q = channel quarter
// (channels 4q..4q+3, one float4).  A wave's float4 load then covers 16 positions x 64 B = 1 KB contiguous, and -- the
// point of the map -- the per-position 16x16 contractions run on the fp32 matrix cores without a single shuffle:
//   lin^T[n][pos] = sum_c W[n][c] xn[pos][c]   as v_mfma_f32_16x16x4_f32 with A[i = n][k] = W[n][4q + kk] (per-lane
//   constants), B[k][j = pos] = xn[pos][4q + kk] (the lane's own float4), and the D fragment (column = pos = lane&15,
//   rows 4*(lane>>4) + r) is lin of the lane's own position and own four channels;
//   g^T[c][pos]  = sum_n W[n][c] d_lin[pos][n] likewise with A[i = c][k] = W[4q + kk][c], accumulated onto the gate term.
// Exact fp32 (these are the fp32 cores: no operand splitting).  The matrix pipe runs beside the VALU, which was the
// bottleneck of the previous register-matvec version (64 + 64 FMAs and 24 shuffles per lane and position); with the
// 64 weight registers gone the weight-gradient outer products fit in the same kernel: ONE backward pass over y.
__device__ float g16_sink[256];

struct f4 { float v[4]; };

__device__ __forceinline__ f4 shfl_xor4(const f4& a, int s) {
  f4 r;
#pragma unroll
  for (int i = 0; i < 4; ++i) r.v[i] = __shfl_xor(a.v[i], s, 64);
  return r;
}

// c += A(16 x 16 per-lane constants a[kk]) * B(16 positions x 16 channels, lane's own b[kk]); operands written by VALU
// code just before are padded away from the MFMAs (see acc_handoff_fence in glu3.hip for the hazard this avoids)
__device__ __forceinline__ f32x4 mm16(const float (&a)[4], float b0, float b1, float b2, float b3, f32x4 c) {
  asm volatile("s_nop 4" : "+v"(b0), "+v"(b1), "+v"(b2), "+v"(b3), "+v"(c));
  c = __builtin_amdgcn_mfma_f32_16x16x4f32(a[0], b0, c, 0, 0, 0);
  c = __builtin_amdgcn_mfma_f32_16x16x4f32(a[1], b1, c, 0, 0, 0);
  c = __builtin_amdgcn_mfma_f32_16x16x4f32(a[2], b2, c, 0, 0, 0);
  c = __builtin_amdgcn_mfma_f32_16x16x4f32(a[3], b3, c, 0, 0, 0);
  return c;
}

__global__ __launch_bounds__(GS_THREADS) void glu16_fwd_kernel(
    const float* __restrict__ y, const float* __restrict__ scale, const float* __restrict__ shift,
    const float* __restrict__ wg, const float* __restrict__ bg, float* __restrict__ out, int B, int H, int W, int ph,
    int pw, float drop_p, uint32_t rng_stream, uint64_t seed) {
  constexpr int C = GS_C;
  const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63, p = lane & 15, q = lane >> 4;
  const int col = wave * 16 + p;
  const int Hp = H / ph, Wp = W / pw;
  float a1[4];  // W[n = p][4q + kk]
#pragma unroll
  for (int kk = 0; kk < 4; ++kk) a1[kk] = wg[p * C + 4 * q + kk];
  float sc[4], sh[4], bi[4];
#pragma unroll
  for (int i = 0; i < 4; ++i) { sc[i] = scale[4 * q + i]; sh[i] = shift[4 * q + i]; bi[i] = bg[4 * q + i]; }
  const uint32_t dkey = drop_key(rng_stream, seed), dthr = drop_threshold16(drop_p);
  const float dscale = drop_p > 0.f ? 1.0f / (1.0f - drop_p) : 1.0f;
  const float inv = 1.0f / (float)(ph * pw);

  for (long item = blockIdx.x; item < (long)B * Hp; item += gridDim.x) {
    const int b = (int)(item / Hp), hp = (int)(item % Hp);
    for (int w0 = 0; w0 < W; w0 += GS_THREADS / 4) {
      const int w = w0 + col;
      const bool ok = w < W;
      const int wc = ok ? w : W - 1;  // clamped: the matrix instructions need every lane, idle columns are masked
      float pooled[4] = {0.f, 0.f, 0.f, 0.f};
      for (int dh = 0; dh < ph; ++dh) {
        const int h = hp * ph + dh;
        const size_t pos = ((size_t)b * H + h) * W + wc;
        const float4 v = *reinterpret_cast<const float4*>(y + pos * C + 4 * q);
        const float xn0 = fmaf(v.x, sc[0], sh[0]), xn1 = fmaf(v.y, sc[1], sh[1]);
        const float xn2 = fmaf(v.z, sc[2], sh[2]), xn3 = fmaf(v.w, sc[3], sh[3]);
        const f32x4 lin = mm16(a1, xn0, xn1, xn2, xn3, f32x4{0.f, 0.f, 0.f, 0.f});
        const float xn[4] = {xn0, xn1, xn2, xn3};
        float dm[4];
        drop_mul2((uint64_t)pos * C + 4 * q, dkey, dthr, dscale, dm[0], dm[1]);
        drop_mul2((uint64_t)pos * C + 4 * q + 2, dkey, dthr, dscale, dm[2], dm[3]);
#pragma unroll
        for (int i = 0; i < 4; ++i) pooled[i] += (lin[i] + bi[i]) * sigmoid_fast(xn[i]) * dm[i];
      }
      if (pw == 2) {
#pragma unroll
        for (int i = 0; i < 4; ++i) pooled[i] += __shfl_xor(pooled[i], 1, 64);  // the neighbouring column
      }
      if (ok && (w & (pw - 1)) == 0 && (w / pw) < Wp) {
        float4 o = make_float4(pooled[0] * inv, pooled[1] * inv, pooled[2] * inv, pooled[3] * inv);
        *reinterpret_cast<float4*>(out + (((size_t)b * Hp + hp) * Wp + w / pw) * C + 4 * q) = o;
      }
    }
  }
}

#define GB16_THREADS 256
// One pass: g, db, the BatchNorm-backward sums AND dW_glu.
// (256 threads: at ~160 VGPRs three 4-wave workgroups fit per CU, one 8-wave workgroup would leave 2 waves per SIMD)
__global__ __launch_bounds__(GB16_THREADS, 3) void glu16_bwd_kernel(
    const float* __restrict__ y, const float* __restrict__ scale, const float* __restrict__ shift,
    const float* __restrict__ wg, const float* __restrict__ bg, const float* __restrict__ dpool,
    float* __restrict__ g_out, float* __restrict__ part_dw /*(G,C,C)*/, float* __restrict__ part_db /*(G,2,C)*/,
    float* __restrict__ part_st /*(G,2,C)*/, int B, int H, int W, int ph, int pw, float drop_p, uint32_t rng_stream,
    uint64_t seed) {
  constexpr int C = GS_C;
  __shared__ float red[GB16_THREADS * 17];
  const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63, p = lane & 15, q = lane >> 4;
  const int col = wave * 16 + p;
  const int Hp = H / ph, Wp = W / pw;
  const int sph = ph >> 1, spw = pw >> 1;
  float a1[4], a2[4];  // W[n = p][4q + kk]  and  W[4q + kk][c = p]
#pragma unroll
  for (int kk = 0; kk < 4; ++kk) { a1[kk] = wg[p * C + 4 * q + kk]; a2[kk] = wg[(4 * q + kk) * C + p]; }
  float sc[4], sh[4], bi[4];
#pragma unroll
  for (int i = 0; i < 4; ++i) { sc[i] = scale[4 * q + i]; sh[i] = shift[4 * q + i]; bi[i] = bg[4 * q + i]; }
  const uint32_t dkey = drop_key(rng_stream, seed), dthr = drop_threshold16(drop_p);
  const float dscale = drop_p > 0.f ? 1.0f / (1.0f - drop_p) : 1.0f;
  const float inv = 1.0f / (float)(ph * pw);

  float dwa[4][4][4];  // dW[4q+i][4(q^s)+kk]: own d_lin rows against the quarter held by lane ^ 16s
  float dba[4] = {0.f, 0.f, 0.f, 0.f}, sga[4] = {0.f, 0.f, 0.f, 0.f}, sgya[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
  for (int s = 0; s < 4; ++s)
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
      for (int kk = 0; kk < 4; ++kk) dwa[s][i][kk] = 0.f;

  // flattened work list: (row of one clip, 128-column chunk); the next item's loads are issued before the current
  // item's arithmetic
  const int chunks = (W + GB16_THREADS / 4 - 1) / (GB16_THREADS / 4);
  const int nitems = B * H * chunks;  // < 2^31 (checked on the host)
  float4 ny = make_float4(0.f, 0.f, 0.f, 0.f), nd = make_float4(0.f, 0.f, 0.f, 0.f);
  auto fetch = [&](int item) {
    const int it = min(item, nitems - 1);  // the last prefetch re-reads the last item: no branch around the loads
    const int ch = it % chunks;
    const int row = it / chunks;
    const int h = row % H;
    const int b = row / H;
    const int w = min(ch * (GB16_THREADS / 4) + col, W - 1);
    const size_t pos = ((size_t)b * H + h) * W + w;
    ny = *reinterpret_cast<const float4*>(y + pos * C + 4 * q);
    const int hp = min(h >> sph, Hp - 1), wpi = min(w >> spw, Wp - 1);
    nd = *reinterpret_cast<const float4*>(dpool + (((size_t)b * Hp + hp) * Wp + wpi) * C + 4 * q);
  };
  fetch(blockIdx.x);
  for (int item = blockIdx.x; item < nitems; item += gridDim.x) {
    const float4 cy = ny, cd = nd;
    fetch(item + (int)gridDim.x);
    const int ch = item % chunks;
    const int row = item / chunks;
    const int h = row % H;
    const int b = row / H;
    const int w = ch * (GB16_THREADS / 4) + col;
    const bool ok = w < W;
    const size_t pos = ((size_t)b * H + h) * W + min(w, W - 1);
    const float okf = ok ? 1.0f : 0.0f;
    const float pmask = (ok && (h >> sph) < Hp && (w >> spw) < Wp) ? inv : 0.f;
    const float yv[4] = {cy.x, cy.y, cy.z, cy.w};
    const float dres[4] = {cd.x, cd.y, cd.z, cd.w};
    f4 xs[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) xs[0].v[i] = fmaf(yv[i], sc[i], sh[i]);
    const f32x4 lin = mm16(a1, xs[0].v[0], xs[0].v[1], xs[0].v[2], xs[0].v[3], f32x4{0.f, 0.f, 0.f, 0.f});
    float dl[4], dm[4];
    f32x4 gt;
    drop_mul2((uint64_t)pos * C + 4 * q, dkey, dthr, dscale, dm[0], dm[1]);
    drop_mul2((uint64_t)pos * C + 4 * q + 2, dkey, dthr, dscale, dm[2], dm[3]);
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const float sg = sigmoid_fast(xs[0].v[i]);
      const float dr = dres[i] * pmask * dm[i];
      dl[i] = dr * sg;
      gt[i] = dr * (lin[i] + bi[i]) * sg * (1.0f - sg);
    }
    const f32x4 g = mm16(a2, dl[0], dl[1], dl[2], dl[3], gt);  // g = d_lin W + gate term
    float* gdst = ok ? g_out + pos * C + 4 * q : g16_sink + 4 * (tid & 63);
    *reinterpret_cast<float4*>(gdst) = make_float4(g[0], g[1], g[2], g[3]);
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      dba[i] += dl[i];
      sga[i] = fmaf(okf, g[i], sga[i]);
      sgya[i] = fmaf(okf * g[i], yv[i], sgya[i]);
    }
    // dW[4q+i][.] += d_lin[i] * xn[.]: the other three quarters of xn come from the lanes 16, 32, 48 apart
    // (idle columns carry d_lin = 0, so their clamped xn contributes nothing)
#pragma unroll
    for (int s = 1; s < 4; ++s) xs[s] = shfl_xor4(xs[0], 16 * s);
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
      for (int s = 0; s < 4; ++s)
#pragma unroll
        for (int kk = 0; kk < 4; ++kk) dwa[s][i][kk] = fmaf(dl[i], xs[s].v[kk], dwa[s][i][kk]);
  }

  // workgroup reduction over the 128 threads that share a quarter index q, 16 values per pass
#pragma unroll
  for (int pass = 0; pass < 5; ++pass) {
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
      for (int wv = 0; wv < GB16_THREADS / 64; ++wv)
        for (int pp = 0; pp < 16; ++pp) s += red[(wv * 64 + 16 * rq + pp) * 17 + j];
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
  BSED_CHECK_ARG((long)B * H * ((W + 63) / 64) + G < (1L << 31), "bsed_glu16_bwd: too many rows");
  hipLaunchKernelGGL(glu16_bwd_kernel, dim3(G), dim3(GB16_THREADS), 0, (hipStream_t)stream, y, scale, shift, wg, bg,
                     dpool, g_out, part_dw, part_db, part_st, B, H, W, ph, pw, drop_p, rng_stream, seed);
  BSED_LAUNCH_CHECK();
  return BSED_OK;
}
