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

// ---------------------------------------------------------------------------------------------------------------
// Matrix-core recurrence (bsed_gru_fwd3 / bsed_gru_bwd3): the default in the split-fp32 contraction mode.
//
// The register/LDS-broadcast kernels above spend ~6500 cycles per time step issuing broadcast ds_read_b128s.  Here
// h @ W_hh^T of every step runs on v_mfma_f32_16x16x32_bf16 with split-fp32 operands (hi = bf16(x), lo = bf16(x-hi);
// hi*hi + hi*lo + lo*hi, fp32 accumulate).  The step is latency-bound, so the layout minimises the serial chain of
// ONE workgroup and spreads the batch over as many CUs as possible:
//   * a workgroup owns only FOUR batch rows of one direction (B = 256 -> 128 workgroups).  The 16-row A tile holds,
//     for batch row b, h_hi in row 4b and h_lo in row 4b+1 (rows 4b+2, 4b+3 are zero), so ONE MFMA against W_hi yields
//     h_hi*W_hi and h_lo*W_hi in registers 0 and 1 of the SAME lane and one more against W_lo yields h_hi*W_lo:
//     two MFMAs per K step instead of three, and the sum is lane-local;
//   * 8 waves; wave w owns hidden units 16w..16w+15 of all three gates, lane (lane>>4, lane&15) = (batch row, unit):
//     the C fragments of the wave's three column tiles hold r, z and n of that (row, unit) -> gate math is lane-local,
//     one element per lane, h stays in a register;
//   * W_hh B-fragments (hi and lo) stay in 96 VGPRs for the whole sequence;
//   * h_t goes through LDS as bf16 rows (272 B pitch = 17 x 16 B: conflict-free 16-byte A-fragment reads),
//     double-buffered -> ONE barrier per step;
//   * the time loop has no divergent branch around memory operations (idle rows load a clamped row and store to a
//     sink) and prefetches the x-projections / saved gates TWO steps ahead, so `s_waitcnt vmcnt(N)` never waits for the
//     step's own stores and HBM latency (~1 us) is covered by two ~0.6 us steps.
typedef short bf16x8 __attribute__((ext_vector_type(8)));
#define GM_RB 4
#define GM_THREADS 512
#define GM_HROW 136  // ushorts per A row of the forward kernel: 128 + 8 pad
#define GM_DROW 392  // ... of the backward kernel: 384 + 8 pad (784 B = 49 x 16 B)

__device__ __forceinline__ unsigned short g_f2bf(float x) {
  const uint32_t u = __float_as_uint(x);
  return (unsigned short)((u + 0x7FFFu + ((u >> 16) & 1u)) >> 16);
}
__device__ __forceinline__ float g_bf2f(unsigned short h) { return __uint_as_float((uint32_t)h << 16); }
__device__ __forceinline__ void g_split8(const float* v, bf16x8& hi, bf16x8& lo) {
#pragma unroll
  for (int j = 0; j < 8; ++j) {
    const unsigned short h = g_f2bf(v[j]);
    hi[j] = (short)h;
    lo[j] = (short)g_f2bf(v[j] - g_bf2f(h));
  }
}
__device__ __forceinline__ float tanh_fast(float x) { return fmaf(2.0f, sigmoid_fast(2.0f * x), -1.0f); }
__device__ float gru_sink[GM_THREADS + 3 * GRU_H];

template <bool SAVE>
__global__ __launch_bounds__(GM_THREADS) void gru_fwd_mfma_kernel(
    const float* __restrict__ xp, const float* __restrict__ w_hh, const float* __restrict__ b_hh,
    float* __restrict__ out, float* __restrict__ gates, int B, int T) {
  __shared__ __align__(16) unsigned short hs[2][16][GM_HROW];
  const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63, lr = lane & 15, lq = lane >> 4;
  const int dir = blockIdx.y;
  const int b0 = blockIdx.x * GM_RB;
  const int u = 16 * wave + lr;

  bf16x8 wh[3][4], wl[3][4];
  float bh[3];
#pragma unroll
  for (int g = 0; g < 3; ++g) {
    bh[g] = b_hh[dir * GRU_G + g * GRU_H + u];
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) {
      const float* p = w_hh + ((size_t)dir * GRU_G + g * GRU_H + u) * GRU_H + 32 * ks + 8 * lq;
      float v[8];
      const float4 v0 = *reinterpret_cast<const float4*>(p), v1 = *reinterpret_cast<const float4*>(p + 4);
      v[0] = v0.x; v[1] = v0.y; v[2] = v0.z; v[3] = v0.w; v[4] = v1.x; v[5] = v1.y; v[6] = v1.z; v[7] = v1.w;
      g_split8(v, wh[g][ks], wl[g][ks]);
    }
  }
  for (int i = tid; i < 2 * 16 * GM_HROW; i += GM_THREADS) (&hs[0][0][0])[i] = 0;

  const bool rok = b0 + lq < B;
  const size_t rbase = (size_t)min(b0 + lq, B - 1) * T;
  const float* xrow = xp + rbase * 768 + dir * GRU_G + u;
  float hold = 0.f;
  float xq[2][3];
  auto load_x = [&](int slot, int sc) {  // sc: step index already clamped to [0, T-1]
    const float* p = xrow + (size_t)(dir == 0 ? sc : T - 1 - sc) * 768;
    xq[slot][0] = p[0]; xq[slot][1] = p[GRU_H]; xq[slot][2] = p[2 * GRU_H];
  };
  load_x(0, 0);
  load_x(1, min(1, T - 1));
  __syncthreads();

  for (int s0 = 0; s0 < T; s0 += 2) {
#pragma unroll
    for (int j = 0; j < 2; ++j) {
      const int s = s0 + j;
      const bool live = s < T;  // T odd: the last half-iteration is a dummy whose stores go to the sink
      const int sc = min(s, T - 1);
      const int t = dir == 0 ? sc : T - 1 - sc;
      bf16x8 a[4];
      const unsigned short* arow = &hs[j][lr][8 * lq];
#pragma unroll
      for (int ks = 0; ks < 4; ++ks) a[ks] = *reinterpret_cast<const bf16x8*>(arow + 32 * ks);
      const float cx0 = xq[j][0], cx1 = xq[j][1], cx2 = xq[j][2];
      load_x(j, min(s + 2, T - 1));
      f32x4 acc1[3], acc2[3];
#pragma unroll
      for (int g = 0; g < 3; ++g) { acc1[g] = f32x4{0.f, 0.f, 0.f, 0.f}; acc2[g] = f32x4{0.f, 0.f, 0.f, 0.f}; }
#pragma unroll
      for (int ks = 0; ks < 4; ++ks)
#pragma unroll
        for (int g = 0; g < 3; ++g) {
          acc1[g] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[ks], wh[g][ks], acc1[g], 0, 0, 0);
          acc2[g] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[ks], wl[g][ks], acc2[g], 0, 0, 0);
        }
      const float ghr = (acc2[0][0] + acc1[0][1]) + acc1[0][0] + bh[0];
      const float ghz = (acc2[1][0] + acc1[1][1]) + acc1[1][0] + bh[1];
      const float ghn = (acc2[2][0] + acc1[2][1]) + acc1[2][0] + bh[2];
      const float rr = sigmoid_fast(cx0 + ghr);
      const float zz = sigmoid_fast(cx1 + ghz);
      // explicit fmas: the SAVE and no-SAVE instantiations must round identically
      const float nn = tanh_fast(fmaf(rr, ghn, cx2));
      const float hnew = fmaf(zz, hold - nn, nn);  // (1-z)*n + z*h
      hold = hnew;
      const unsigned short hi = g_f2bf(hnew);
      hs[j ^ 1][4 * lq][u] = hi;
      hs[j ^ 1][4 * lq + 1][u] = g_f2bf(hnew - g_bf2f(hi));
      const bool ok = live && rok;
      const size_t bt = rbase + t;
      float* po = ok ? out + bt * 256 + dir * GRU_H + u : gru_sink + tid;
      *po = hnew;
      if (SAVE) {
        float* gp = ok ? gates + ((bt * 2 + dir) * 4) * GRU_H + u : gru_sink + tid;
        gp[0] = rr; gp[GRU_H] = zz; gp[2 * GRU_H] = nn; gp[3 * GRU_H] = ghn;
      }
      __syncthreads();
    }
  }
}

__global__ __launch_bounds__(GM_THREADS) void gru_bwd_mfma_kernel(
    const float* __restrict__ dout, const float* __restrict__ out, const float* __restrict__ gates,
    const float* __restrict__ w_hh, float* __restrict__ dxp, float* __restrict__ dgh,
    float* __restrict__ part_bih, float* __restrict__ part_bhh, int B, int T) {
  __shared__ __align__(16) unsigned short ds[2][16][GM_DROW];
  const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63, lr = lane & 15, lq = lane >> 4;
  const int dir = blockIdx.y;
  const int b0 = blockIdx.x * GM_RB;
  const int u = 16 * wave + lr;

  // dh_prev[b][u] += sum_n dg[b][n] W_hh[n][u]: B fragment = W_hh[n = 32ks + 8lq + j][u]
  bf16x8 wh[12], wl[12];
#pragma unroll
  for (int ks = 0; ks < 12; ++ks) {
    float v[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) v[j] = w_hh[((size_t)dir * GRU_G + 32 * ks + 8 * lq + j) * GRU_H + u];
    g_split8(v, wh[ks], wl[ks]);
  }
  for (int i = tid; i < 2 * 16 * GM_DROW; i += GM_THREADS) (&ds[0][0][0])[i] = 0;
  const bool rok = b0 + lq < B;
  const size_t rbase = (size_t)min(b0 + lq, B - 1) * T;
  float dhrec = 0.f;
  float sum_r = 0.f, sum_z = 0.f, sum_n = 0.f, sum_g = 0.f;  // bias gradients: sums over time of this (row, unit)
  float q_do[2], q_r[2], q_z[2], q_n[2], q_ghn[2], q_hp[2];
  auto load_step = [&](int slot, int sc) {  // sc clamped to [0, T-1]; branch-free
    const int t = dir == 0 ? sc : T - 1 - sc;
    const int tp = min(max(dir == 0 ? t - 1 : t + 1, 0), T - 1);
    const size_t bt = rbase + t;
    q_do[slot] = dout[bt * 256 + dir * GRU_H + u];
    const float* gp = gates + ((bt * 2 + dir) * 4) * GRU_H + u;
    q_r[slot] = gp[0]; q_z[slot] = gp[GRU_H]; q_n[slot] = gp[2 * GRU_H]; q_ghn[slot] = gp[3 * GRU_H];
    const float hp = out[(rbase + tp) * 256 + dir * GRU_H + u];
    q_hp[slot] = sc > 0 ? hp : 0.f;
  };
  load_step(0, T - 1);
  load_step(1, max(T - 2, 0));
  __syncthreads();

  for (int s0 = T - 1; s0 >= 0; s0 -= 2) {
#pragma unroll
    for (int j = 0; j < 2; ++j) {
      const int s = s0 - j;
      const bool live = s >= 0;
      const int sc = max(s, 0);
      const int t = dir == 0 ? sc : T - 1 - sc;
      const float dh = q_do[j] + dhrec;
      const float rr = q_r[j], zz = q_z[j], nn = q_n[j], ghn = q_ghn[j], hp = q_hp[j];
      load_step(j, max(s - 2, 0));
      const float dn_pre = dh * (1.0f - zz) * (1.0f - nn * nn);
      const float dz_pre = dh * (hp - nn) * zz * (1.0f - zz);
      const float dr_pre = dn_pre * ghn * rr * (1.0f - rr);
      const float dghn = dn_pre * rr;
      const float dd = dh * zz;
      unsigned short* dhi = &ds[j][4 * lq][0];
      unsigned short* dlo = &ds[j][4 * lq + 1][0];
      const unsigned short h0 = g_f2bf(dr_pre), h1 = g_f2bf(dz_pre), h2 = g_f2bf(dghn);
      dhi[u] = h0; dhi[GRU_H + u] = h1; dhi[2 * GRU_H + u] = h2;
      dlo[u] = g_f2bf(dr_pre - g_bf2f(h0));
      dlo[GRU_H + u] = g_f2bf(dz_pre - g_bf2f(h1));
      dlo[2 * GRU_H + u] = g_f2bf(dghn - g_bf2f(h2));
      const bool ok = live && rok;
      const float okf = ok ? 1.0f : 0.0f;
      sum_r = fmaf(okf, dr_pre, sum_r); sum_z = fmaf(okf, dz_pre, sum_z);
      sum_n = fmaf(okf, dn_pre, sum_n); sum_g = fmaf(okf, dghn, sum_g);
      const size_t o = (rbase + t) * 768 + dir * GRU_G + u;
      float* px = ok ? dxp + o : gru_sink + tid;
      float* ph = ok ? dgh + o : gru_sink + tid;
      px[0] = dr_pre; px[GRU_H] = dz_pre; px[2 * GRU_H] = dn_pre;
      ph[0] = dr_pre; ph[GRU_H] = dz_pre; ph[2 * GRU_H] = dghn;
      __syncthreads();
      f32x4 acc1 = f32x4{0.f, 0.f, 0.f, 0.f}, acc2 = f32x4{0.f, 0.f, 0.f, 0.f};
      const unsigned short* arow = &ds[j][lr][8 * lq];
#pragma unroll
      for (int ks = 0; ks < 12; ++ks) {
        const bf16x8 a = *reinterpret_cast<const bf16x8*>(arow + 32 * ks);
        acc1 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, wh[ks], acc1, 0, 0, 0);
        acc2 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, wl[ks], acc2, 0, 0, 0);
      }
      dhrec = dd + ((acc2[0] + acc1[1]) + acc1[0]);
      // the other half-iteration writes ds[j ^ 1]; ds[j] is rewritten two steps later, after a barrier every wave
      // reaches only once its reads above are done: one barrier per step
    }
  }
  if (part_bih) {  // per batch-row partials of db_ih = [dr, dz, dn] and db_hh = [dr, dz, d(W_hn h + b_hn)]
    const size_t row = (size_t)blockIdx.x * GM_RB + lq;
    float* pi = part_bih + row * 768 + dir * GRU_G + u;
    float* ph = part_bhh + row * 768 + dir * GRU_G + u;
    pi[0] = sum_r; pi[GRU_H] = sum_z; pi[2 * GRU_H] = sum_n;
    ph[0] = sum_r; ph[GRU_H] = sum_z; ph[2 * GRU_H] = sum_g;
  }
}

extern "C" int bsed_gru_fwd(const float* xp, const float* w_hh, const float* b_hh, float* out, float* gates, int B,
                            int T, int rows_per_wg, void* stream) {
  BSED_CHECK_ARG(xp && w_hh && b_hh && out, "bsed_gru_fwd: null tensor");
  BSED_CHECK_ARG(B > 0 && T > 0 && rows_per_wg > 0, "bsed_gru_fwd: bad shape");
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
  BSED_CHECK_ARG(B > 0 && T > 0 && rows_per_wg > 0, "bsed_gru_bwd: bad shape");
  hipStream_t s = (hipStream_t)stream;
  const int R = rows_per_wg;
  dim3 grid(ceil_div(B, R), 2);
  if (R == 1) hipLaunchKernelGGL(gru_bwd_kernel<1>, grid, dim3(GRU_THREADS), 0, s, dout, out, gates, w_hh, dxp, dgh, B, T);
  else if (R == 2) hipLaunchKernelGGL(gru_bwd_kernel<2>, grid, dim3(GRU_THREADS), 0, s, dout, out, gates, w_hh, dxp, dgh, B, T);
  else { bsed_set_error("bsed_gru_bwd: rows_per_wg must be 1 or 2"); return BSED_ERR_ARG; }
  BSED_LAUNCH_CHECK();
  return BSED_OK;
}

extern "C" int bsed_gru_fwd3(const float* xp, const float* w_hh, const float* b_hh, float* out, float* gates, int B,
                             int T, void* stream) {
  BSED_CHECK_ARG(xp && w_hh && b_hh && out, "bsed_gru_fwd3: null tensor");
  BSED_CHECK_ARG(B > 0 && T > 0, "bsed_gru_fwd3: bad shape");
  hipStream_t s = (hipStream_t)stream;
  dim3 grid(ceil_div(B, GM_RB), 2);
  if (gates) hipLaunchKernelGGL(gru_fwd_mfma_kernel<true>, grid, dim3(GM_THREADS), 0, s, xp, w_hh, b_hh, out, gates, B, T);
  else hipLaunchKernelGGL(gru_fwd_mfma_kernel<false>, grid, dim3(GM_THREADS), 0, s, xp, w_hh, b_hh, out, gates, B, T);
  BSED_LAUNCH_CHECK();
  return BSED_OK;
}

extern "C" int bsed_gru_bwd3_rows(int B) { return ceil_div(B, GM_RB) * GM_RB; }

extern "C" int bsed_gru_bwd3(const float* dout, const float* out, const float* gates, const float* w_hh, float* dxp,
                             float* dgh, float* part_bih, float* part_bhh, int B, int T, void* stream) {
  BSED_CHECK_ARG(dout && out && gates && w_hh && dxp && dgh, "bsed_gru_bwd3: null tensor");
  BSED_CHECK_ARG((part_bih == nullptr) == (part_bhh == nullptr), "bsed_gru_bwd3: part_bih and part_bhh come together");
  BSED_CHECK_ARG(B > 0 && T > 0, "bsed_gru_bwd3: bad shape");
  dim3 grid(ceil_div(B, GM_RB), 2);
  hipLaunchKernelGGL(gru_bwd_mfma_kernel, grid, dim3(GM_THREADS), 0, (hipStream_t)stream, dout, out, gates, w_hh, dxp,
                     dgh, part_bih, part_bhh, B, T);
  BSED_LAUNCH_CHECK();
  return BSED_OK;
}
