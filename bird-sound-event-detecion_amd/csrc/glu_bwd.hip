// Fused backward of the GLU stage for C = 32 / 64 / 128 channels: ONE pass over the pre-BN activation y.
//
// Reference math (src/models/CNN.py:5-16,59-67, autograd of BN-apply -> Linear -> sigmoid gate -> Dropout -> AvgPool):
//   xn = y*scale + shift                      (BatchNorm apply)
//   lin = xn W^T + b,  sig = sigmoid(xn),  res = lin*sig,  pooled = avgpool(dropout(res))
//   d_res = unpool(d_pooled) * mask/(1-p) / window
//   d_lin = d_res*sig                         -> db += sum d_lin,  dW += d_lin^T xn
//   g     = d_lin W + d_res*lin*sig*(1-sig)   = dL/d xn, plus the BatchNorm-backward sums (sum g, sum g*y)
// The unfused chain (three launches + a weight-gradient launch) moved ~9.5 activation-sized tensors through HBM;
// here y is read once (plus an L2-hot re-read), g is written once, and d_lin / the gate term never leave the chip:
//   GEMM1 (lin)  : A = xn tile in LDS,   B = W^T slabs           -> accumulators
//   epilogue 1   : d_lin -> LDS,  gate term stays in the SAME accumulators (they seed GEMM2)
//   GEMM2 (g)    : A = d_lin tile in LDS, B = W slabs
//   GEMM3 (dW)   : A = d_lin^T, B = xn, K = the tile's 128 positions, accumulators persistent over the tiles of a
//                  workgroup; one partial slab per workgroup at the end (deterministic, no float atomics)
// All contractions on v_mfma_f32_32x32x2_f32; lane maps as in igemm.hip.
#include "bsed_common.h"
#include "../../include/bsed.h"
#include <algorithm>

#define GB_THREADS 256
#define GB_M 128

struct GluBwdParams {
  const float* y; const float* scale; const float* shift;
  const float* wfwd;   // [c][n] = W[n][c]   (GEMM1 operand)
  const float* wbwd;   // [n][c] = W[n][c]   (GEMM2 operand; the PyTorch tensor itself)
  const float* bias; const float* dpool;
  float* g; float* part_dw; float* part_db; float* part_st;
  int NB, H, W, TH, TW, lgTW, tilesH, tilesW, ntiles;
  int ph, pw, Hp, Wp;
  float drop_p; uint32_t rng_stream; uint64_t seed;
};

__device__ __forceinline__ int crow2(int r, int lh) { return (r & 3) + 8 * (r >> 2) + 4 * lh; }

// NW = 8 (C = 128 only): the eight waves split the channel tiles in two halves, so every SIMD hosts two waves and one
// wave's staging / epilogue latency hides behind the other's MFMAs (the 148 KB of LDS allow a single workgroup per CU).
template <int C, int NW>
__global__ __launch_bounds__(NW * 64) void glu_bwd_fused_kernel(const GluBwdParams P) {
  constexpr int NTHR = NW * 64;
  constexpr int NTA = C / 32;                    // 32-wide channel tiles in total
  constexpr int NT = NTA / (NW / 4);             // ... per wave
  constexpr int XP = C + 1;
  constexpr int KSPLIT = NTA >= 4 ? 1 : 4 / NTA; // waves that share one dW row-tile split the positions
  constexpr bool RES = C <= 32;                  // both weight matrices stay resident in LDS: no slab traffic / barriers
  extern __shared__ __align__(16) float smem[];
  float* Xs = smem;                 // [128][XP]  BatchNorm output
  float* Ds = Xs + GB_M * XP;       // [128][XP]  d_lin
  float* Bs = Ds + GB_M * XP;       // [32][C]    weight slab   (RES: [2][C][C] = W^T then W, loaded once)
  const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63, li = lane & 31, lh = lane >> 5;
  const int rbase = (wave & 3) * 32;             // the 32 tile rows (positions) of this wave
  const int cbase = (wave >> 2) * NT * 32;       // first channel of this wave's column tiles
  const int m = rbase + li;
  const int sph = P.ph >> 1, spw = P.pw >> 1;
  const float inv_pool = 1.0f / (float)(P.ph * P.pw);
  const uint32_t dkey = drop_key(P.rng_stream, P.seed), dthr = drop_threshold(P.drop_p);
  const float dscale = P.drop_p > 0.f ? 1.0f / (1.0f - P.drop_p) : 1.0f;

  float bias[NT], sdb[NT], sgs[NT], sgy[NT];
  f32x16 acc3[NT];
#pragma unroll
  for (int j = 0; j < NT; ++j) {
    bias[j] = P.bias[cbase + 32 * j + li];
    sdb[j] = 0.f; sgs[j] = 0.f; sgy[j] = 0.f;
#pragma unroll
    for (int r = 0; r < 16; ++r) acc3[j][r] = 0.f;
  }
  const int nt3 = NW == 8 ? (wave & 3) : wave % NTA, kq = NW == 8 ? 0 : wave / NTA;  // GEMM3 role of this wave
  constexpr int KR = GB_M / KSPLIT;
  if (RES) {
    for (int e = tid; e < C * C / 4; e += NTHR) {
      reinterpret_cast<float4*>(Bs)[e] = reinterpret_cast<const float4*>(P.wfwd)[e];
      reinterpret_cast<float4*>(Bs + C * C)[e] = reinterpret_cast<const float4*>(P.wbwd)[e];
    }
  }

  for (int tile0 = blockIdx.x; tile0 < P.ntiles; tile0 += gridDim.x) {
    int tile = tile0;
    const int tw_i = tile % P.tilesW; tile /= P.tilesW;
    const int th_i = tile % P.tilesH;
    const int nb = tile / P.tilesH;
    const int th0 = th_i * P.TH, tw0 = tw_i * P.TW;
    __syncthreads();  // every wave is done with Xs/Ds of the previous tile (GEMM3)
    // ---- stage xn = y*scale + shift
    constexpr int C4 = C / 4;
    for (int e0 = tid; e0 < GB_M * C4; e0 += 4 * NTHR) {
      float4 v[4];
      bool okv[4];
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        const int e = e0 + u * NTHR;
        const int c4 = e % C4, mm = e / C4;
        const int gh = th0 + (mm >> P.lgTW), gw = tw0 + (mm & (P.TW - 1));
        okv[u] = gh < P.H && gw < P.W;
        v[u] = make_float4(0.f, 0.f, 0.f, 0.f);
        if (okv[u]) v[u] = *reinterpret_cast<const float4*>(P.y + (((size_t)nb * P.H + gh) * P.W + gw) * C + 4 * c4);
      }
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        const int e = e0 + u * NTHR;
        const int c4 = e % C4, mm = e / C4;
        if (okv[u]) {
          const float4 sc = *reinterpret_cast<const float4*>(P.scale + 4 * c4);
          const float4 sh = *reinterpret_cast<const float4*>(P.shift + 4 * c4);
          v[u].x = fmaf(v[u].x, sc.x, sh.x); v[u].y = fmaf(v[u].y, sc.y, sh.y);
          v[u].z = fmaf(v[u].z, sc.z, sh.z); v[u].w = fmaf(v[u].w, sc.w, sh.w);
        }
        float* dst = Xs + mm * XP + 4 * c4;
        dst[0] = v[u].x; dst[1] = v[u].y; dst[2] = v[u].z; dst[3] = v[u].w;
      }
    }
    f32x16 acc[NT];
#pragma unroll
    for (int j = 0; j < NT; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[j][r] = 0.f;

    // ---- GEMM1: lin = xn W^T
    if (RES) __syncthreads();  // staged tile (and, first time, the resident weights) visible
    for (int k0 = 0; k0 < C; k0 += 32) {
      if (!RES) {
        __syncthreads();
        for (int e = tid; e < 32 * C4; e += NTHR) {
          const int k = e / C4, n4 = e % C4;
          *reinterpret_cast<float4*>(Bs + k * C + 4 * n4) = *reinterpret_cast<const float4*>(P.wfwd + (size_t)(k0 + k) * C + 4 * n4);
        }
        __syncthreads();
      }
      const float* arow = Xs + m * XP + k0 + lh;
      const float* brow = (RES ? Bs + k0 * C : Bs) + lh * C + cbase + li;
#pragma unroll 4
      for (int kk = 0; kk < 32; kk += 2) {
        const float a = arow[kk];
#pragma unroll
        for (int j = 0; j < NT; ++j) acc[j] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, brow[kk * C + 32 * j], acc[j], 0, 0, 0);
      }
    }

    // ---- epilogue 1: d_lin -> LDS, gate term -> accumulators
#pragma unroll
    for (int rg = 0; rg < 4; ++rg) {
      float dv[4][NT];
      int mmv[4];
      bool okr[4];
      size_t posv[4];
#pragma unroll
      for (int rr = 0; rr < 4; ++rr) {
        const int mm = rbase + crow2(rg * 4 + rr, lh);
        const int gh = th0 + (mm >> P.lgTW), gw = tw0 + (mm & (P.TW - 1));
        const bool pok = gh < P.H && gw < P.W;
        const int gph = gh >> sph, gpw = gw >> spw;
        const bool pooled_ok = pok && gph < P.Hp && gpw < P.Wp;
        mmv[rr] = mm; okr[rr] = pok;
        posv[rr] = ((size_t)nb * P.H + gh) * P.W + gw;
        const float* dprow = P.dpool + (((size_t)nb * P.Hp + gph) * P.Wp + gpw) * C + cbase + li;
#pragma unroll
        for (int j = 0; j < NT; ++j) dv[rr][j] = pooled_ok ? dprow[32 * j] : 0.f;
      }
#pragma unroll
      for (int rr = 0; rr < 4; ++rr) {
        const int r = rg * 4 + rr;
#pragma unroll
        for (int j = 0; j < NT; ++j) {
          const int n = cbase + 32 * j + li;
          float dl = 0.f, tt = 0.f;
          if (okr[rr]) {
            const float xn = Xs[mmv[rr] * XP + n];
            const float sg = sigmoid_fast(xn);
            const float lin = acc[j][r] + bias[j];
            const float dres = dv[rr][j] * inv_pool * drop_mul((uint64_t)posv[rr] * C + n, dkey, dthr, dscale);
            dl = dres * sg;
            tt = dres * lin * sg * (1.0f - sg);
            sdb[j] += dl;
          }
          Ds[mmv[rr] * XP + n] = dl;
          acc[j][r] = tt;
        }
      }
    }

    // (accumulators just written by VALU code become SrcC of the next MFMA: finish the copies into the tuples and
    // pad the distance -- see acc_handoff_fence in glu3.hip)
#pragma unroll
    for (int j = 0; j < NT; ++j) asm volatile("s_nop 7" : "+v"(acc[j]));
    // ---- GEMM2: g = d_lin W + gate term (already in acc); its A rows are this wave's own d_lin rows
    for (int k0 = 0; k0 < C; k0 += 32) {
      if (!RES) {
        __syncthreads();
        for (int e = tid; e < 32 * C4; e += NTHR) {
          const int k = e / C4, n4 = e % C4;
          *reinterpret_cast<float4*>(Bs + k * C + 4 * n4) = *reinterpret_cast<const float4*>(P.wbwd + (size_t)(k0 + k) * C + 4 * n4);
        }
        __syncthreads();
      }
      const float* arow = Ds + m * XP + k0 + lh;
      const float* brow = (RES ? Bs + C * C + k0 * C : Bs) + lh * C + cbase + li;
#pragma unroll 4
      for (int kk = 0; kk < 32; kk += 2) {
        const float a = arow[kk];
#pragma unroll
        for (int j = 0; j < NT; ++j) acc[j] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, brow[kk * C + 32 * j], acc[j], 0, 0, 0);
      }
    }

    // ---- epilogue 2: write g, BatchNorm-backward sums (y re-read is L2-hot)
#pragma unroll
    for (int rg = 0; rg < 4; ++rg) {
      float yv[4][NT];
      bool okr[4];
      size_t posv[4];
#pragma unroll
      for (int rr = 0; rr < 4; ++rr) {
        const int mm = rbase + crow2(rg * 4 + rr, lh);
        const int gh = th0 + (mm >> P.lgTW), gw = tw0 + (mm & (P.TW - 1));
        okr[rr] = gh < P.H && gw < P.W;
        posv[rr] = ((size_t)nb * P.H + gh) * P.W + gw;
#pragma unroll
        for (int j = 0; j < NT; ++j) yv[rr][j] = okr[rr] ? P.y[posv[rr] * C + cbase + 32 * j + li] : 0.f;
      }
#pragma unroll
      for (int rr = 0; rr < 4; ++rr) {
        const int r = rg * 4 + rr;
        if (okr[rr]) {
#pragma unroll
          for (int j = 0; j < NT; ++j) {
            const float gv = acc[j][r];
            P.g[posv[rr] * C + cbase + 32 * j + li] = gv;
            sgs[j] += gv;
            sgy[j] = fmaf(gv, yv[rr][j], sgy[j]);
          }
        }
      }
    }

    // ---- GEMM3: dW[n][c] += sum_p d_lin[p][n] xn[p][c]   (needs every wave's d_lin rows)
    if (RES) __syncthreads();
    {
      const float* abase = Ds + nt3 * 32 + li;
      const float* bbase = Xs + cbase + li;
#pragma unroll 2
      for (int kp = kq * KR; kp < (kq + 1) * KR; kp += 2) {
        const int mk = kp + lh;
        const float a = abase[mk * XP];
        float bv[NT];
#pragma unroll
        for (int j = 0; j < NT; ++j) bv[j] = bbase[mk * XP + 32 * j];
#pragma unroll
        for (int j = 0; j < NT; ++j) acc3[j] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, bv[j], acc3[j], 0, 0, 0);
      }
    }
  }

  // ---- per-workgroup partials
#pragma unroll
  for (int j = 0; j < NT; ++j)
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int n = nt3 * 32 + crow2(r, lh);
      P.part_dw[(((size_t)blockIdx.x * KSPLIT + kq) * C + n) * C + cbase + 32 * j + li] = acc3[j][r];
    }
  __syncthreads();
  float* red = smem;  // [4 row groups][3][C]; wave (row group, half) owns the columns of its half
#pragma unroll
  for (int j = 0; j < NT; ++j) {
    const float a = sdb[j] + __shfl_xor(sdb[j], 32, 64);
    const float b = sgs[j] + __shfl_xor(sgs[j], 32, 64);
    const float c = sgy[j] + __shfl_xor(sgy[j], 32, 64);
    if (lh == 0) {
      red[((wave & 3) * 3 + 0) * C + cbase + 32 * j + li] = a;
      red[((wave & 3) * 3 + 1) * C + cbase + 32 * j + li] = b;
      red[((wave & 3) * 3 + 2) * C + cbase + 32 * j + li] = c;
    }
  }
  __syncthreads();
  for (int e = tid; e < 3 * C; e += NTHR) {
    const int which = e / C, n = e % C;
    const float s = red[(0 * 3 + which) * C + n] + red[(1 * 3 + which) * C + n] + red[(2 * 3 + which) * C + n] +
                    red[(3 * 3 + which) * C + n];
    if (which == 0) {
      P.part_db[((size_t)blockIdx.x * 2 + 0) * C + n] = s;
      P.part_db[((size_t)blockIdx.x * 2 + 1) * C + n] = 0.f;
    } else {
      P.part_st[((size_t)blockIdx.x * 2 + (which - 1)) * C + n] = s;
    }
  }
}

template <int C, int NW>
static int launch_glu_bwd(const GluBwdParams& P, int G, hipStream_t s) {
  const size_t smem = ((size_t)2 * GB_M * (C + 1) + (C <= 32 ? 2 * C * C : 32 * C)) * sizeof(float);
  static BsedLdsOnce once;
  BSED_HIP(bsed_max_lds(once, (const void*)glu_bwd_fused_kernel<C, NW>));
  hipLaunchKernelGGL((glu_bwd_fused_kernel<C, NW>), dim3(G), dim3(NW * 64), smem, s, P);
  BSED_LAUNCH_CHECK();
  return BSED_OK;
}

extern "C" int bsed_glu_bwd_slabs(int C) { return C >= 128 ? 1 : (C == 64 ? 2 : 4); }

extern "C" int bsed_glu_bwd_fused(const float* y, const float* scale, const float* shift, const float* wfwd,
                                  const float* wbwd, const float* bias, const float* dpool, float* g, float* part_dw,
                                  float* part_db, float* part_st, int G, int NB, int H, int W, int C, int TH, int TW,
                                  int ph, int pw, float drop_p, uint32_t rng_stream, uint64_t seed, void* stream) {
  BSED_CHECK_ARG(y && scale && shift && wfwd && wbwd && bias && dpool && g && part_dw && part_db && part_st,
                 "bsed_glu_bwd_fused: null tensor");
  BSED_CHECK_ARG(C == 32 || C == 64 || C == 128, "bsed_glu_bwd_fused: built for C in {32,64,128} (got %d)", C);
  BSED_CHECK_ARG(NB > 0 && H > 0 && W > 0 && G > 0 && TH * TW == GB_M && W % TW == 0, "bsed_glu_bwd_fused: bad shape");
  BSED_CHECK_ARG((ph == 1 || ph == 2) && (pw == 1 || pw == 2), "bsed_glu_bwd_fused: pooling windows must be 1 or 2");
  GluBwdParams P;
  P.y = y; P.scale = scale; P.shift = shift; P.wfwd = wfwd; P.wbwd = wbwd; P.bias = bias; P.dpool = dpool;
  P.g = g; P.part_dw = part_dw; P.part_db = part_db; P.part_st = part_st;
  P.NB = NB; P.H = H; P.W = W; P.TH = TH; P.TW = TW;
  P.lgTW = 0;
  while ((1 << P.lgTW) < TW) ++P.lgTW;
  BSED_CHECK_ARG((1 << P.lgTW) == TW, "bsed_glu_bwd_fused: TW must be a power of two");
  P.tilesH = ceil_div(H, TH); P.tilesW = W / TW;
  const long ntiles = (long)NB * P.tilesH * P.tilesW;
  BSED_CHECK_ARG(ntiles < (1L << 31) && G <= ntiles, "bsed_glu_bwd_fused: G must not exceed the %ld tiles", ntiles);
  P.ntiles = (int)ntiles;
  P.ph = ph; P.pw = pw; P.Hp = H / ph; P.Wp = W / pw;
  P.drop_p = drop_p; P.rng_stream = rng_stream; P.seed = seed;
  hipStream_t s = (hipStream_t)stream;
  if (C == 128) return launch_glu_bwd<128, 8>(P, G, s);
  if (C == 64) return launch_glu_bwd<64, 4>(P, G, s);
  return launch_glu_bwd<32, 4>(P, G, s);
}
