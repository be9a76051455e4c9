// Implicit-GEMM family on v_mfma_f32_32x32x2_f32 (exact fp32 matrix cores of gfx950).
//
//   igemm_kernel : out[p][n] = sum_{tap,k} in[p + off(tap)][k] * w[tap][k][n]   (NHWC, zero padding)
//                  M = 128 output positions of a TH x TW spatial tile, N = 32/64/128 channels,
//                  K = taps x CIN, staged through LDS in (tap, 32-channel) slabs.
//                  One kernel body serves (reference call sites in brackets):
//                    conv 3x3 forward + BatchNorm batch statistics   [src/models/CNN.py:46-49]
//                    conv 3x3 data gradient (flipped taps)           [autograd of the above]
//                    GLU: BN-apply on load, 1x1 contraction, sigmoid gate, dropout, avg-pool
//                         fused in the epilogue                      [src/models/CNN.py:5-16,59-67]
//                    GLU backward pre/post stages
//                    GRU input projections / their data gradients    [src/models/RNN.py:12 (nn.GRU)]
//   wgrad_kernel : dW[tap][k][n] = sum_p in[p + off(tap)][k] * dy[p][n]  (K of the GEMM = positions),
//                  persistent over position tiles, partial slabs reduced by reduce_partials_kernel.
//
// Lane maps (pinned on hardware by bsed_selftest_mfma): A[i=lane&31][k=lane>>5],
// B[k=lane>>5][j=lane&31], C[row=(r&3)+8*(r>>2)+4*(lane>>5)][col=lane&31].
#include "bsed_common.h"
#include "../../include/bsed.h"
#include <algorithm>

#define IG_THREADS 256
#define IG_TILE_M 128

enum { EPI_PLAIN = 0, EPI_STATS = 1, EPI_GLU_POOL = 2, EPI_GLU_BWD = 3, EPI_ADD_STATS2 = 4 };

struct IgemmParams {
  BsedIgemmDesc d;
  int PW, PH, PP, lgTW, b_off;  // derived on the host
};

__device__ __forceinline__ int crow(int r, int lh) { return (r & 3) + 8 * (r >> 2) + 4 * lh; }

template <int KC, int BN, int EPI>
__global__ __launch_bounds__(IG_THREADS) void igemm_kernel(const IgemmParams P) {
  constexpr int NT = BN / 32;
  constexpr int AP = KC + 1;
  const BsedIgemmDesc& p = P.d;
  extern __shared__ __align__(16) float smem[];
  float* As = smem;
  float* Bs = smem + P.b_off;
  const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63, li = lane & 31, lh = lane >> 5;
  int tile = blockIdx.x;
  const int tw_i = tile % p.tilesW; tile /= p.tilesW;
  const int th_i = tile % p.tilesH;
  const int nb = tile / p.tilesH;
  const int th0 = th_i * p.TH, tw0 = tw_i * p.TW;
  const int n0 = blockIdx.y * BN;
  const int PW = P.PW;
  const int m = wave * 32 + li;
  const int abase = (((m >> P.lgTW) + p.hh) * PW + (m & (p.TW - 1)) + p.hw) * AP;

  f32x16 acc[NT];
#pragma unroll
  for (int j = 0; j < NT; ++j)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[j][r] = 0.f;

  const float* inb = p.in + (size_t)nb * p.H * p.W * p.in_pitch;
  for (int c0 = 0; c0 < p.CIN; c0 += KC) {
    __syncthreads();
    for (int e = tid; e < P.PP * (KC / 4); e += IG_THREADS) {
      const int c4 = e % (KC / 4), pos = e / (KC / 4);
      const int pr = pos / PW, pc = pos - pr * PW;
      const int gh = th0 - p.hh + pr, gw = tw0 - p.hw + pc;
      float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
      if (gh >= 0 && gh < p.H && gw >= 0 && gw < p.W) {
        v = *reinterpret_cast<const float4*>(inb + ((size_t)gh * p.W + gw) * p.in_pitch + c0 + 4 * c4);
        if (p.a_scale) {
          const float4 sc = *reinterpret_cast<const float4*>(p.a_scale + c0 + 4 * c4);
          const float4 sh = *reinterpret_cast<const float4*>(p.a_shift + c0 + 4 * c4);
          v.x = fmaf(v.x, sc.x, sh.x); v.y = fmaf(v.y, sc.y, sh.y);
          v.z = fmaf(v.z, sc.z, sh.z); v.w = fmaf(v.w, sc.w, sh.w);
        }
      }
      float* dst = As + pos * AP + 4 * c4;
      dst[0] = v.x; dst[1] = v.y; dst[2] = v.z; dst[3] = v.w;
    }
    for (int tap = 0; tap < p.ntaps; ++tap) {
      if (tap > 0) __syncthreads();
      const float* wsrc = p.w + ((size_t)tap * p.CIN + c0) * p.NP + n0;
      for (int e = tid; e < KC * BN / 4; e += IG_THREADS) {
        const int k = e / (BN / 4), n4 = e % (BN / 4);
        *reinterpret_cast<float4*>(Bs + k * BN + 4 * n4) =
            *reinterpret_cast<const float4*>(wsrc + (size_t)k * p.NP + 4 * n4);
      }
      __syncthreads();
      const float* arow = As + abase + (p.dh[tap] * PW + p.dw[tap]) * AP + lh;
      const float* brow = Bs + lh * BN + li;
#pragma unroll 4
      for (int kk = 0; kk < KC; kk += 2) {
        const float a = arow[kk];
#pragma unroll
        for (int j = 0; j < NT; ++j) {
          const float b = brow[kk * BN + 32 * j];
          acc[j] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc[j], 0, 0, 0);
        }
      }
    }
  }

  // ------------------------------------------------------------------------------------ epilogue
  float s0[NT], s1[NT];
#pragma unroll
  for (int j = 0; j < NT; ++j) { s0[j] = 0.f; s1[j] = 0.f; }
  if (EPI == EPI_GLU_POOL) __syncthreads();  // As/Bs are recycled as the pooling stage
  float* Cs = smem;                          // [128][BN+1]

#pragma unroll
  for (int j = 0; j < NT; ++j) {
    const int n = n0 + 32 * j + li;
    const bool nok = n < p.N;
    const float bias = (p.bias && nok) ? p.bias[n] : 0.f;
    float esc = 0.f, esh = 0.f;
    if (EPI == EPI_GLU_POOL || EPI == EPI_GLU_BWD) {
      if (nok) { esc = p.e_scale[n]; esh = p.e_shift[n]; }
    }
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int mm = wave * 32 + crow(r, lh);
      const int gh = th0 + (mm >> P.lgTW), gw = tw0 + (mm & (p.TW - 1));
      const bool ok = nok && gh < p.H && gw < p.W;
      const size_t pos = ((size_t)nb * p.H + gh) * p.W + gw;
      float v = acc[j][r] + bias;
      if (EPI == EPI_PLAIN) {
        if (ok) p.out[pos * p.out_pitch + n] = v;
      } else if (EPI == EPI_STATS) {
        if (ok) {
          p.out[pos * p.out_pitch + n] = v;
          s0[j] += v;
          s1[j] = fmaf(v, v, s1[j]);
        }
      } else if (EPI == EPI_GLU_POOL) {
        float res = 0.f;
        if (ok) {
          const float xn = fmaf(p.e_src[pos * p.e_pitch + n], esc, esh);
          res = v * sigmoidf_(xn) * dropout_scale(pos * p.N + n, p.rng_stream, p.seed, p.drop_p);
        }
        Cs[mm * (BN + 1) + 32 * j + li] = res;
      } else if (EPI == EPI_GLU_BWD) {
        if (ok) {
          const float xn = fmaf(p.e_src[pos * p.e_pitch + n], esc, esh);
          const float sg = sigmoidf_(xn);
          float dres = 0.f;
          const int gph = gh / p.ph, gpw = gw / p.pw;
          if (gph < p.Hp && gpw < p.Wp) {
            dres = p.e_dpool[(((size_t)nb * p.Hp + gph) * p.Wp + gpw) * p.N + n] * (1.0f / (float)(p.ph * p.pw)) *
                   dropout_scale(pos * p.N + n, p.rng_stream, p.seed, p.drop_p);
          }
          const float dlin = dres * sg;
          p.out[pos * p.out_pitch + n] = dlin;
          p.out2[pos * p.out_pitch + n] = dres * v * sg * (1.0f - sg);
          s0[j] += dlin;
        }
      } else {  // EPI_ADD_STATS2: g = acc + residual ; stats = (sum g, sum g*y)
        if (ok) {
          const float g = v + p.out2[pos * p.out_pitch + n];
          p.out[pos * p.out_pitch + n] = g;
          s0[j] += g;
          s1[j] = fmaf(g, p.e_src[pos * p.e_pitch + n], s1[j]);
        }
      }
    }
  }

  if (EPI == EPI_GLU_POOL) {
    __syncthreads();
    const int tpw = p.TW / p.pw, tph = p.TH / p.ph;
    const float inv = 1.0f / (float)(p.ph * p.pw);
    for (int e = tid; e < tph * tpw * BN; e += IG_THREADS) {
      const int n = e % BN, pp = e / BN;
      const int pr = pp / tpw, pc = pp - pr * tpw;
      const int gph = th0 / p.ph + pr, gpw = tw0 / p.pw + pc;
      if (gph < p.Hp && gpw < p.Wp && n0 + n < p.N) {
        float s = 0.f;
        for (int i = 0; i < p.ph; ++i)
          for (int jj = 0; jj < p.pw; ++jj)
            s += Cs[((pr * p.ph + i) * p.TW + pc * p.pw + jj) * (BN + 1) + n];
        p.out[(((size_t)nb * p.Hp + gph) * p.Wp + gpw) * p.out_pitch + n0 + n] = s * inv;
      }
    }
  }

  if (EPI == EPI_STATS || EPI == EPI_GLU_BWD || EPI == EPI_ADD_STATS2) {
    __syncthreads();
    float* red = smem;  // [4][2][BN]
#pragma unroll
    for (int j = 0; j < NT; ++j) {
      const float a = s0[j] + __shfl_xor(s0[j], 32, 64);
      const float b = s1[j] + __shfl_xor(s1[j], 32, 64);
      if (lh == 0) {
        red[(wave * 2 + 0) * BN + 32 * j + li] = a;
        red[(wave * 2 + 1) * BN + 32 * j + li] = b;
      }
    }
    __syncthreads();
    if (tid < 2 * BN) {
      const int which = tid / BN, n = tid % BN;
      if (n0 + n < p.N) {
        const float s = red[(0 * 2 + which) * BN + n] + red[(1 * 2 + which) * BN + n] +
                        red[(2 * 2 + which) * BN + n] + red[(3 * 2 + which) * BN + n];
        p.stats[((size_t)blockIdx.x * 2 + which) * p.N + n0 + n] = s;
      }
    }
  }
}

// ---------------------------------------------------------------------------------------------
// weight-gradient kernel
// ---------------------------------------------------------------------------------------------
struct WgradParams {
  BsedWgradDesc d;
  int PW, PH, PP, lgTW, dy_off, ntiles, nct;  // nct = CINP/32
};

template <int MAXS>
__global__ __launch_bounds__(IG_THREADS) void wgrad_kernel(const WgradParams P) {
  const BsedWgradDesc& p = P.d;
  extern __shared__ __align__(16) float smem[];
  const int XP = p.CINP + 1;
  float* Xs = smem;
  float* DYs = smem + P.dy_off;  // [128][32]
  const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63, li = lane & 31, lh = lane >> 5;
  const int n0 = blockIdx.y * 32;
  const int PW = P.PW;
  const int nitems = p.ntaps * P.nct;

  int xoff[MAXS];
  bool valid[MAXS];
  f32x16 acc[MAXS];
#pragma unroll
  for (int s = 0; s < MAXS; ++s) {
    const int it = wave + 4 * s;
    valid[s] = it < nitems;
    const int tap = valid[s] ? it / P.nct : 0, cit = valid[s] ? it % P.nct : 0;
    xoff[s] = (p.dh[tap] * PW + p.dw[tap]) * XP + cit * 32 + li;
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[s][r] = 0.f;
  }

  for (int tile0 = blockIdx.x; tile0 < P.ntiles; tile0 += gridDim.x) {
    int tile = tile0;
    const int tw_i = tile % p.tilesW; tile /= p.tilesW;
    const int th_i = tile % p.tilesH;
    const int nb = tile / p.tilesH;
    const int th0 = th_i * p.TH, tw0 = tw_i * p.TW;
    const float* inb = p.in + (size_t)nb * p.H * p.W * p.in_pitch;
    const float* dyb = p.dy + (size_t)nb * p.H * p.W * p.dy_pitch;
    __syncthreads();
    const int c4n = p.CINP / 4;
    for (int e = tid; e < P.PP * c4n; e += IG_THREADS) {
      const int c4 = e % c4n, pos = e / c4n;
      const int pr = pos / PW, pc = pos - pr * PW;
      const int gh = th0 - p.hh + pr, gw = tw0 - p.hw + pc;
      float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
      if (gh >= 0 && gh < p.H && gw >= 0 && gw < p.W && 4 * c4 < p.CIN) {
        v = *reinterpret_cast<const float4*>(inb + ((size_t)gh * p.W + gw) * p.in_pitch + 4 * c4);
        if (p.a_scale) {
          const float4 sc = *reinterpret_cast<const float4*>(p.a_scale + 4 * c4);
          const float4 sh = *reinterpret_cast<const float4*>(p.a_shift + 4 * c4);
          v.x = fmaf(v.x, sc.x, sh.x); v.y = fmaf(v.y, sc.y, sh.y);
          v.z = fmaf(v.z, sc.z, sh.z); v.w = fmaf(v.w, sc.w, sh.w);
        }
      }
      float* dst = Xs + pos * XP + 4 * c4;
      dst[0] = v.x; dst[1] = v.y; dst[2] = v.z; dst[3] = v.w;
    }
    for (int e = tid; e < IG_TILE_M * 8; e += IG_THREADS) {
      const int n4 = e & 7, mm = e >> 3;
      const int gh = th0 + (mm >> P.lgTW), gw = tw0 + (mm & (p.TW - 1));
      float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
      if (gh < p.H && gw < p.W && n0 + 4 * n4 < p.N)
        v = *reinterpret_cast<const float4*>(dyb + ((size_t)gh * p.W + gw) * p.dy_pitch + n0 + 4 * n4);
      *reinterpret_cast<float4*>(DYs + mm * 32 + 4 * n4) = v;
    }
    __syncthreads();
#pragma unroll 2
    for (int kp = 0; kp < IG_TILE_M; kp += 2) {
      const int mk = kp + lh;
      const float b = DYs[mk * 32 + li];
      const float* xrow = Xs + (((mk >> P.lgTW) + p.hh) * PW + (mk & (p.TW - 1)) + p.hw) * XP;
#pragma unroll
      for (int s = 0; s < MAXS; ++s) {
        if (valid[s]) {
          const float a = xrow[xoff[s]];
          acc[s] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc[s], 0, 0, 0);
        }
      }
    }
  }
  const int NPo = gridDim.y * 32;
#pragma unroll
  for (int s = 0; s < MAXS; ++s) {
    if (valid[s]) {
      const int it = wave + 4 * s;
      const int tap = it / P.nct, cit = it % P.nct;
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int ci = cit * 32 + crow(r, lh);
        p.part[(((size_t)blockIdx.x * p.ntaps + tap) * p.CINP + ci) * NPo + n0 + li] = acc[s][r];
      }
    }
  }
}

// dst[tap*s_tap + k*s_k + n*s_n] (+)= sum_g part[g][tap][k][n]
__global__ void reduce_partials_kernel(const float* __restrict__ part, int G, int ntaps, int KP, int NP, int K,
                                       int N, float* __restrict__ dst, long s_tap, long s_k, long s_n,
                                       int accumulate) {
  const long total = (long)ntaps * KP * NP;
  for (long e = (long)blockIdx.x * blockDim.x + threadIdx.x; e < total; e += (long)gridDim.x * blockDim.x) {
    const int n = (int)(e % NP);
    const long r = e / NP;
    const int k = (int)(r % KP), tap = (int)(r / KP);
    if (n >= N || k >= K) continue;
    float s = 0.f;
    for (int g = 0; g < G; ++g) s += part[(size_t)g * total + e];
    float* d = dst + tap * s_tap + k * s_k + n * s_n;
    *d = accumulate ? *d + s : s;
  }
}

// wpk[tap][k][n] = src[tap*s_tap + k*s_k + n*s_n], zero for n >= N
__global__ void pack_weight_kernel(const float* __restrict__ src, float* __restrict__ dst, int ntaps, int K, int N,
                                   int NP, long s_tap, long s_k, long s_n) {
  const long total = (long)ntaps * K * NP;
  for (long e = (long)blockIdx.x * blockDim.x + threadIdx.x; e < total; e += (long)gridDim.x * blockDim.x) {
    const int n = (int)(e % NP);
    const long r = e / NP;
    const int k = (int)(r % K), tap = (int)(r / K);
    dst[e] = n < N ? src[tap * s_tap + k * s_k + n * s_n] : 0.f;
  }
}

// ---------------------------------------------------------------------------------------------
// host side
// ---------------------------------------------------------------------------------------------
static int ilog2_exact(int v) {
  int l = 0;
  while ((1 << l) < v) ++l;
  return (1 << l) == v ? l : -1;
}

template <int KC, int BN>
static int launch_igemm_epi(const IgemmParams& P, dim3 grid, size_t smem, hipStream_t s) {
  switch (P.d.epilogue) {
#define CASE(E)                                                                                       \
  case E: {                                                                                           \
    static bool done = false;                                                                         \
    if (!done) {                                                                                      \
      BSED_HIP(hipFuncSetAttribute((const void*)igemm_kernel<KC, BN, E>,                              \
                                   hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));          \
      done = true;                                                                                    \
    }                                                                                                 \
    hipLaunchKernelGGL((igemm_kernel<KC, BN, E>), grid, dim3(IG_THREADS), smem, s, P);                \
    break;                                                                                            \
  }
    CASE(EPI_PLAIN)
    CASE(EPI_STATS)
    CASE(EPI_GLU_POOL)
    CASE(EPI_GLU_BWD)
    CASE(EPI_ADD_STATS2)
#undef CASE
    default:
      bsed_set_error("bsed_igemm: unknown epilogue %d", P.d.epilogue);
      return BSED_ERR_ARG;
  }
  BSED_LAUNCH_CHECK();
  return BSED_OK;
}

extern "C" int bsed_igemm_num_tiles(const BsedIgemmDesc* d) {
  if (!d || d->TH <= 0 || d->TW <= 0) return -1;
  return d->NB * ceil_div(d->H, d->TH) * ceil_div(d->W, d->TW);
}

extern "C" int bsed_igemm(const BsedIgemmDesc* desc, void* stream) {
  BSED_CHECK_ARG(desc, "bsed_igemm: null descriptor");
  IgemmParams P;
  P.d = *desc;
  BsedIgemmDesc& d = P.d;
  BSED_CHECK_ARG(d.in && d.w && d.out, "bsed_igemm: null tensor");
  BSED_CHECK_ARG(d.NB > 0 && d.H > 0 && d.W > 0 && d.CIN > 0 && d.N > 0, "bsed_igemm: bad shape");
  BSED_CHECK_ARG(d.TH * d.TW == IG_TILE_M, "bsed_igemm: TH*TW must be 128 (got %dx%d)", d.TH, d.TW);
  P.lgTW = ilog2_exact(d.TW);
  BSED_CHECK_ARG(P.lgTW >= 0, "bsed_igemm: TW must be a power of two");
  BSED_CHECK_ARG(d.W % d.TW == 0, "bsed_igemm: W %% TW != 0");
  BSED_CHECK_ARG(d.ntaps >= 1 && d.ntaps <= 9, "bsed_igemm: ntaps must be in 1..9");
  for (int t = 0; t < d.ntaps; ++t)
    BSED_CHECK_ARG(abs(d.dh[t]) <= d.hh && abs(d.dw[t]) <= d.hw, "bsed_igemm: tap %d outside the halo", t);
  const int KC = (d.CIN % 32 == 0) ? 32 : 16;
  BSED_CHECK_ARG(d.CIN % KC == 0, "bsed_igemm: CIN must be a multiple of 16");
  BSED_CHECK_ARG(d.in_pitch >= d.CIN && d.in_pitch % 4 == 0 && d.out_pitch >= d.N, "bsed_igemm: bad pitch");
  BSED_CHECK_ARG(d.NP % 32 == 0 && d.NP >= d.N, "bsed_igemm: NP must be N rounded up to 32");
  const int BN = d.NP % 128 == 0 ? 128 : (d.NP % 64 == 0 ? 64 : 32);
  d.tilesH = ceil_div(d.H, d.TH);
  d.tilesW = d.W / d.TW;
  P.PW = d.TW + 2 * d.hw;
  P.PH = d.TH + 2 * d.hh;
  P.PP = P.PW * P.PH;
  P.b_off = (P.PP * (KC + 1) + 3) & ~3;
  size_t fl = (size_t)P.b_off + (size_t)KC * BN;
  if (d.epilogue == EPI_GLU_POOL) {
    BSED_CHECK_ARG(d.ph >= 1 && d.pw >= 1 && d.TH % d.ph == 0 && d.TW % d.pw == 0, "bsed_igemm: tile not pool aligned");
    BSED_CHECK_ARG(d.Hp == d.H / d.ph && d.Wp == d.W / d.pw, "bsed_igemm: bad pooled shape");
    fl = std::max(fl, (size_t)IG_TILE_M * (BN + 1));
  }
  if (d.epilogue == EPI_GLU_POOL || d.epilogue == EPI_GLU_BWD)
    BSED_CHECK_ARG(d.e_src && d.e_scale && d.e_shift && d.e_pitch >= d.N, "bsed_igemm: GLU epilogue needs e_src/e_scale/e_shift");
  if (d.epilogue == EPI_GLU_BWD)
    BSED_CHECK_ARG(d.e_dpool && d.out2 && d.stats && d.ph >= 1 && d.pw >= 1 && d.Hp == d.H / d.ph && d.Wp == d.W / d.pw,
                   "bsed_igemm: GLU_BWD epilogue needs e_dpool/out2/stats and pooled shape");
  if (d.epilogue == EPI_ADD_STATS2)
    BSED_CHECK_ARG(d.out2 && d.e_src && d.stats && d.e_pitch >= d.N, "bsed_igemm: ADD_STATS2 needs out2/e_src/stats");
  if (d.epilogue == EPI_STATS) BSED_CHECK_ARG(d.stats, "bsed_igemm: STATS epilogue needs a stats buffer");
  fl = std::max(fl, (size_t)8 * BN);
  const size_t smem = fl * sizeof(float);
  BSED_CHECK_ARG(smem <= 160 * 1024, "bsed_igemm: tile needs %zu B of LDS", smem);
  const long ntiles = (long)d.NB * d.tilesH * d.tilesW;
  BSED_CHECK_ARG(ntiles < (1L << 31), "bsed_igemm: too many tiles");
  dim3 grid((unsigned)ntiles, d.NP / BN);
  hipStream_t s = (hipStream_t)stream;
  if (KC == 32) {
    if (BN == 128) return launch_igemm_epi<32, 128>(P, grid, smem, s);
    if (BN == 64) return launch_igemm_epi<32, 64>(P, grid, smem, s);
    return launch_igemm_epi<32, 32>(P, grid, smem, s);
  }
  if (BN == 128) return launch_igemm_epi<16, 128>(P, grid, smem, s);
  if (BN == 64) return launch_igemm_epi<16, 64>(P, grid, smem, s);
  return launch_igemm_epi<16, 32>(P, grid, smem, s);
}

template <int MAXS>
static int launch_wgrad(const WgradParams& P, dim3 grid, size_t smem, hipStream_t s) {
  static bool done = false;
  if (!done) {
    BSED_HIP(hipFuncSetAttribute((const void*)wgrad_kernel<MAXS>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
    done = true;
  }
  hipLaunchKernelGGL((wgrad_kernel<MAXS>), grid, dim3(IG_THREADS), smem, s, P);
  BSED_LAUNCH_CHECK();
  return BSED_OK;
}

extern "C" int bsed_wgrad(const BsedWgradDesc* desc, void* stream) {
  BSED_CHECK_ARG(desc, "bsed_wgrad: null descriptor");
  WgradParams P;
  P.d = *desc;
  BsedWgradDesc& d = P.d;
  BSED_CHECK_ARG(d.in && d.dy && d.part, "bsed_wgrad: null tensor");
  BSED_CHECK_ARG(d.NB > 0 && d.H > 0 && d.W > 0 && d.CIN > 0 && d.N > 0 && d.G > 0, "bsed_wgrad: bad shape");
  BSED_CHECK_ARG(d.TH * d.TW == IG_TILE_M, "bsed_wgrad: TH*TW must be 128");
  P.lgTW = ilog2_exact(d.TW);
  BSED_CHECK_ARG(P.lgTW >= 0 && d.W % d.TW == 0, "bsed_wgrad: TW must be a power of two dividing W");
  BSED_CHECK_ARG(d.ntaps >= 1 && d.ntaps <= 9, "bsed_wgrad: ntaps must be in 1..9");
  for (int t = 0; t < d.ntaps; ++t)
    BSED_CHECK_ARG(abs(d.dh[t]) <= d.hh && abs(d.dw[t]) <= d.hw, "bsed_wgrad: tap %d outside the halo", t);
  BSED_CHECK_ARG(d.CIN % 4 == 0 && d.N % 4 == 0, "bsed_wgrad: CIN and N must be multiples of 4");
  BSED_CHECK_ARG(d.CINP % 32 == 0 && d.CINP >= d.CIN && d.NP % 32 == 0 && d.NP >= d.N, "bsed_wgrad: bad padding");
  BSED_CHECK_ARG(d.in_pitch >= d.CIN && d.in_pitch % 4 == 0 && d.dy_pitch >= d.N && d.dy_pitch % 4 == 0, "bsed_wgrad: bad pitch");
  d.tilesH = ceil_div(d.H, d.TH);
  d.tilesW = d.W / d.TW;
  P.PW = d.TW + 2 * d.hw;
  P.PH = d.TH + 2 * d.hh;
  P.PP = P.PW * P.PH;
  P.nct = d.CINP / 32;
  P.dy_off = (P.PP * (d.CINP + 1) + 3) & ~3;
  const size_t smem = ((size_t)P.dy_off + IG_TILE_M * 32) * sizeof(float);
  BSED_CHECK_ARG(smem <= 160 * 1024, "bsed_wgrad: tile needs %zu B of LDS", smem);
  const long ntiles = (long)d.NB * d.tilesH * d.tilesW;
  BSED_CHECK_ARG(ntiles < (1L << 31), "bsed_wgrad: too many tiles");
  P.ntiles = (int)ntiles;
  const int nitems = d.ntaps * P.nct;
  const int slots = ceil_div(nitems, 4);
  dim3 grid((unsigned)std::min<long>(d.G, ntiles), d.NP / 32);
  BSED_CHECK_ARG((int)grid.x == d.G, "bsed_wgrad: G (%d) exceeds the number of tiles (%ld)", d.G, ntiles);
  hipStream_t s = (hipStream_t)stream;
  if (slots <= 1) return launch_wgrad<1>(P, grid, smem, s);
  if (slots <= 2) return launch_wgrad<2>(P, grid, smem, s);
  if (slots <= 3) return launch_wgrad<3>(P, grid, smem, s);
  if (slots <= 5) return launch_wgrad<5>(P, grid, smem, s);
  if (slots <= 9) return launch_wgrad<9>(P, grid, smem, s);
  bsed_set_error("bsed_wgrad: %d work items per workgroup exceed the 36 supported", nitems);
  return BSED_ERR_ARG;
}

extern "C" int bsed_reduce_partials(const float* part, int G, int ntaps, int KP, int NP, int K, int N, float* dst,
                                    long s_tap, long s_k, long s_n, int accumulate, void* stream) {
  BSED_CHECK_ARG(part && dst && G > 0 && ntaps > 0 && KP >= K && NP >= N && K > 0 && N > 0, "bsed_reduce_partials: bad argument");
  const long total = (long)ntaps * KP * NP;
  hipLaunchKernelGGL(reduce_partials_kernel, dim3((unsigned)std::min<long>(ceil_div(total, 256), 4096)), dim3(256), 0,
                     (hipStream_t)stream, part, G, ntaps, KP, NP, K, N, dst, s_tap, s_k, s_n, accumulate);
  BSED_LAUNCH_CHECK();
  return BSED_OK;
}

extern "C" int bsed_pack_weight(const float* src, float* dst, int ntaps, int K, int N, int NP, long s_tap, long s_k,
                                long s_n, void* stream) {
  BSED_CHECK_ARG(src && dst && ntaps > 0 && K > 0 && N > 0 && NP >= N, "bsed_pack_weight: bad argument");
  const long total = (long)ntaps * K * NP;
  hipLaunchKernelGGL(pack_weight_kernel, dim3((unsigned)std::min<long>(ceil_div(total, 256), 4096)), dim3(256), 0,
                     (hipStream_t)stream, src, dst, ntaps, K, N, NP, s_tap, s_k, s_n);
  BSED_LAUNCH_CHECK();
  return BSED_OK;
}
