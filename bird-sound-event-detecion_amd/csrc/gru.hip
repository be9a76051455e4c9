// Bidirectional GRU recurrence (nn.GRU semantics, gate order r,z,n, b_hn inside the r-gated term).
// Replaces the cuDNN/ATen RNN the reference reaches through src/models/RNN.py:7-16.
//
// The input projections x @ W_ih^T + b_ih for ALL time steps are one MFMA GEMM (igemm.hip); what is
// left is the strictly sequential part, 313 steps of h @ W_hh^T (128 x 384) plus the gate math.
// That part is latency-bound, so it is laid out for the shortest possible step rather than for MFMA:
//   * one workgroup = R batch rows of one direction (R=2 at B=256 fills all 256 CUs: 128 x 2 WGs);
//   * W_hh lives in REGISTERS for the whole sequence: thread j owns gate row j (128 VGPRs);
//   * h_t is broadcast from LDS (ds_read_b128, every lane the same address -> no bank conflicts);
//   * two workgroup barriers per step; the x-projection loads for step t+1 are issued before the
//     matvec of step t so their HBM latency is hidden.
// The backward kernel mirrors it with W_hh columns in registers (thread (k,q): column k of gate q)
// and emits the pre-activation gradients the weight/data GEMMs consume afterwards.
#include "bsed_common.h"
#include "../../include/bsed.h"

#define GRU_H 128
#define GRU_G 384
#define GRU_THREADS 384

template <int R>
__global__ __launch_bounds__(GRU_THREADS) void gru_fwd_kernel(
    const float* __restrict__ xp,    // (B,T,768): [dir*384 + gate*128 + k], b_ih already added
    const float* __restrict__ w_hh,  // (2,384,128)
    const float* __restrict__ b_hh,  // (2,384)
    float* __restrict__ out,         // (B,T,256): [dir*128 + k]
    float* __restrict__ gates,       // (B,T,2,4,128) r,z,n,gh_n  (nullable: inference)
    int B, int T) {
  __shared__ __align__(16) float hs[R][GRU_H];
  __shared__ float gh[R][GRU_G];
  const int tid = threadIdx.x;
  const int dir = blockIdx.y;
  const int b0 = blockIdx.x * R;

  float w[GRU_H];
  {
    const float4* wr = reinterpret_cast<const float4*>(w_hh + ((size_t)dir * GRU_G + tid) * GRU_H);
#pragma unroll
    for (int k = 0; k < GRU_H / 4; ++k) {
      const float4 v = wr[k];
      w[4 * k] = v.x; w[4 * k + 1] = v.y; w[4 * k + 2] = v.z; w[4 * k + 3] = v.w;
    }
  }
  const float bh = b_hh[dir * GRU_G + tid];
  for (int i = tid; i < R * GRU_H; i += GRU_THREADS) (&hs[0][0])[i] = 0.f;

  // combine-phase role: thread -> (row r, unit k)
  const bool comb = tid < R * GRU_H;
  const int cr = tid / GRU_H, ck = tid % GRU_H;
  const int cb = b0 + cr;
  const bool cok = comb && cb < B;
  float xr = 0.f, xz = 0.f, xn = 0.f;
  auto load_x = [&](int t) {
    if (cok) {
      const float* p = xp + ((size_t)cb * T + t) * 768 + dir * GRU_G + ck;
      xr = p[0]; xz = p[GRU_H]; xn = p[2 * GRU_H];
    }
  };
  load_x(dir == 0 ? 0 : T - 1);
  __syncthreads();

  for (int s = 0; s < T; ++s) {
    const int t = dir == 0 ? s : T - 1 - s;
    const float cxr = xr, cxz = xz, cxn = xn;
    if (s + 1 < T) load_x(dir == 0 ? s + 1 : T - 2 - s);  // prefetch the next step
    float a[R];
#pragma unroll
    for (int r = 0; r < R; ++r) a[r] = bh;
#pragma unroll
    for (int k = 0; k < GRU_H; k += 4) {
#pragma unroll
      for (int r = 0; r < R; ++r) {
        const float4 h4 = *reinterpret_cast<const float4*>(&hs[r][k]);
        a[r] = fmaf(w[k], h4.x, a[r]);
        a[r] = fmaf(w[k + 1], h4.y, a[r]);
        a[r] = fmaf(w[k + 2], h4.z, a[r]);
        a[r] = fmaf(w[k + 3], h4.w, a[r]);
      }
    }
#pragma unroll
    for (int r = 0; r < R; ++r) gh[r][tid] = a[r];
    __syncthreads();
    if (comb) {
      const float rr = sigmoidf_(cxr + gh[cr][ck]);
      const float zz = sigmoidf_(cxz + gh[cr][GRU_H + ck]);
      const float ghn = gh[cr][2 * GRU_H + ck];
      const float nn = tanhf(cxn + rr * ghn);
      const float hold = hs[cr][ck];
      const float hnew = (1.0f - zz) * nn + zz * hold;
      hs[cr][ck] = hnew;
      if (cok) {
        out[((size_t)cb * T + t) * 256 + dir * GRU_H + ck] = hnew;
        if (gates) {
          float* g = gates + ((((size_t)cb * T + t) * 2 + dir) * 4) * GRU_H + ck;
          g[0] = rr; g[GRU_H] = zz; g[2 * GRU_H] = nn; g[3 * GRU_H] = ghn;
        }
      }
    }
    __syncthreads();
  }
}

template <int R>
__global__ __launch_bounds__(GRU_THREADS) void gru_bwd_kernel(
    const float* __restrict__ dout,   // (B,T,256) gradient w.r.t. the layer output
    const float* __restrict__ out,    // (B,T,256) forward output (h_t)
    const float* __restrict__ gates,  // (B,T,2,4,128)
    const float* __restrict__ w_hh,   // (2,384,128)
    float* __restrict__ dxp,          // (B,T,768) input-side pre-activation gradients
    float* __restrict__ dgh,          // (B,T,768) hidden-side pre-activation gradients
    int B, int T) {
  __shared__ __align__(16) float dg[R][GRU_G];
  __shared__ float part[R][3][GRU_H];
  const int tid = threadIdx.x;
  const int dir = blockIdx.y;
  const int b0 = blockIdx.x * R;
  const int k = tid & (GRU_H - 1), q = tid >> 7;  // column k of gate block q

  float w[GRU_H];
#pragma unroll
  for (int j = 0; j < GRU_H; ++j) w[j] = w_hh[((size_t)dir * GRU_G + q * GRU_H + j) * GRU_H + k];

  const bool comb = tid < R * GRU_H;
  const int cr = tid / GRU_H, ck = tid % GRU_H;
  const int cb = b0 + cr;
  const bool cok = comb && cb < B;
  float dhrec = 0.f;

  float n_do = 0.f, n_r = 0.f, n_z = 0.f, n_n = 0.f, n_ghn = 0.f, n_hp = 0.f;
  auto load_step = [&](int s) {
    if (cok) {
      const int t = dir == 0 ? s : T - 1 - s;
      const size_t bt = (size_t)cb * T + t;
      n_do = dout[bt * 256 + dir * GRU_H + ck];
      const float* g = gates + ((bt * 2 + dir) * 4) * GRU_H + ck;
      n_r = g[0]; n_z = g[GRU_H]; n_n = g[2 * GRU_H]; n_ghn = g[3 * GRU_H];
      n_hp = 0.f;
      if (s > 0) {
        const int tp = dir == 0 ? t - 1 : t + 1;
        n_hp = out[((size_t)cb * T + tp) * 256 + dir * GRU_H + ck];
      }
    }
  };
  load_step(T - 1);

  for (int s = T - 1; s >= 0; --s) {
    const int t = dir == 0 ? s : T - 1 - s;
    float dd = 0.f;
    if (comb) {
      const float dh = n_do + dhrec;
      const float rr = n_r, zz = n_z, nn = n_n, ghn = n_ghn, hp = n_hp;
      const float dn_pre = dh * (1.0f - zz) * (1.0f - nn * nn);
      const float dz_pre = dh * (hp - nn) * zz * (1.0f - zz);
      const float dr_pre = dn_pre * ghn * rr * (1.0f - rr);
      const float dghn = dn_pre * rr;
      dd = dh * zz;
      dg[cr][ck] = dr_pre;
      dg[cr][GRU_H + ck] = dz_pre;
      dg[cr][2 * GRU_H + ck] = dghn;
      if (cok) {
        const size_t o = ((size_t)cb * T + t) * 768 + dir * GRU_G + ck;
        dxp[o] = dr_pre; dxp[o + GRU_H] = dz_pre; dxp[o + 2 * GRU_H] = dn_pre;
        dgh[o] = dr_pre; dgh[o + GRU_H] = dz_pre; dgh[o + 2 * GRU_H] = dghn;
      }
    }
    __syncthreads();
    if (s > 0) load_step(s - 1);  // prefetch while the matvec runs
    float a[R];
#pragma unroll
    for (int r = 0; r < R; ++r) a[r] = 0.f;
#pragma unroll
    for (int j = 0; j < GRU_H; j += 4) {
#pragma unroll
      for (int r = 0; r < R; ++r) {
        const float4 d4 = *reinterpret_cast<const float4*>(&dg[r][q * GRU_H + j]);
        a[r] = fmaf(w[j], d4.x, a[r]);
        a[r] = fmaf(w[j + 1], d4.y, a[r]);
        a[r] = fmaf(w[j + 2], d4.z, a[r]);
        a[r] = fmaf(w[j + 3], d4.w, a[r]);
      }
    }
#pragma unroll
    for (int r = 0; r < R; ++r) part[r][q][k] = a[r];
    __syncthreads();
    if (comb) dhrec = dd + part[cr][0][ck] + part[cr][1][ck] + part[cr][2][ck];
    // next iteration's writes to dg happen after every wave has passed the barrier above, and its
    // writes to part after the barrier that follows them: two barriers per step are enough
  }
}

extern "C" int bsed_gru_fwd(const float* xp, const float* w_hh, const float* b_hh, float* out, float* gates, int B,
                            int T, int rows_per_wg, void* stream) {
  BSED_CHECK_ARG(xp && w_hh && b_hh && out, "bsed_gru_fwd: null tensor");
  BSED_CHECK_ARG(B > 0 && T > 0, "bsed_gru_fwd: bad shape");
  hipStream_t s = (hipStream_t)stream;
  const int R = rows_per_wg;
  dim3 grid(ceil_div(B, R), 2);
  if (R == 1) hipLaunchKernelGGL(gru_fwd_kernel<1>, grid, dim3(GRU_THREADS), 0, s, xp, w_hh, b_hh, out, gates, B, T);
  else if (R == 2) hipLaunchKernelGGL(gru_fwd_kernel<2>, grid, dim3(GRU_THREADS), 0, s, xp, w_hh, b_hh, out, gates, B, T);
  else if (R == 4) hipLaunchKernelGGL(gru_fwd_kernel<4>, grid, dim3(GRU_THREADS), 0, s, xp, w_hh, b_hh, out, gates, B, T);
  else { bsed_set_error("bsed_gru_fwd: rows_per_wg must be 1, 2 or 4"); return BSED_ERR_ARG; }
  BSED_LAUNCH_CHECK();
  return BSED_OK;
}

extern "C" int bsed_gru_bwd(const float* dout, const float* out, const float* gates, const float* w_hh, float* dxp,
                            float* dgh, int B, int T, int rows_per_wg, void* stream) {
  BSED_CHECK_ARG(dout && out && gates && w_hh && dxp && dgh, "bsed_gru_bwd: null tensor");
  BSED_CHECK_ARG(B > 0 && T > 0, "bsed_gru_bwd: bad shape");
  hipStream_t s = (hipStream_t)stream;
  const int R = rows_per_wg;
  dim3 grid(ceil_div(B, R), 2);
  if (R == 1) hipLaunchKernelGGL(gru_bwd_kernel<1>, grid, dim3(GRU_THREADS), 0, s, dout, out, gates, w_hh, dxp, dgh, B, T);
  else if (R == 2) hipLaunchKernelGGL(gru_bwd_kernel<2>, grid, dim3(GRU_THREADS), 0, s, dout, out, gates, w_hh, dxp, dgh, B, T);
  else { bsed_set_error("bsed_gru_bwd: rows_per_wg must be 1 or 2"); return BSED_ERR_ARG; }
  BSED_LAUNCH_CHECK();
  return BSED_OK;
}
