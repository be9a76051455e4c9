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
#include <type_traits>

#define IG_THREADS 256
#define IG_TILE_M 128
#define W3_DY_BYTES (2 * IG_TILE_M * 32 * 2)  // bf16x3 weight gradient: LDS bytes of one 32-channel dy tile (hi + lo planes)

enum { EPI_PLAIN = 0, EPI_STATS = 1, EPI_GLU_POOL = 2, EPI_GLU_BWD = 3, EPI_ADD_STATS2 = 4 };

struct IgemmParams {
  BsedIgemmDesc d;
  int PW, PH, PP, lgTW, b_off, pw_magic;  // derived on the host; pos / PW == (pos * pw_magic) >> 20
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
    // global loads are issued four at a time before any LDS store so their latencies overlap
    const int a_total = P.PP * (KC / 4);
    for (int e0 = tid; e0 < a_total; e0 += 4 * IG_THREADS) {
      float4 v[4];
      bool okv[4];
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        const int e = e0 + u * IG_THREADS;
        const int c4 = e % (KC / 4), pos = e / (KC / 4);
        const int pr = (pos * P.pw_magic) >> 20, pc = pos - pr * PW;
        const int gh = th0 - p.hh + pr, gw = tw0 - p.hw + pc;
        v[u] = make_float4(0.f, 0.f, 0.f, 0.f);
        okv[u] = e < a_total && gh >= 0 && gh < p.H && gw >= 0 && gw < p.W;
        if (okv[u]) v[u] = *reinterpret_cast<const float4*>(inb + ((size_t)gh * p.W + gw) * p.in_pitch + c0 + 4 * c4);
      }
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        const int e = e0 + u * IG_THREADS;
        if (e < a_total) {
          const int c4 = e % (KC / 4), pos = e / (KC / 4);
          if (p.a_scale && okv[u]) {
            const float4 sc = *reinterpret_cast<const float4*>(p.a_scale + c0 + 4 * c4);
            const float4 sh = *reinterpret_cast<const float4*>(p.a_shift + c0 + 4 * c4);
            v[u].x = fmaf(v[u].x, sc.x, sh.x); v[u].y = fmaf(v[u].y, sc.y, sh.y);
            v[u].z = fmaf(v[u].z, sc.z, sh.z); v[u].w = fmaf(v[u].w, sc.w, sh.w);
          }
          float* dst = As + pos * AP + 4 * c4;
          dst[0] = v[u].x; dst[1] = v[u].y; dst[2] = v[u].z; dst[3] = v[u].w;
        }
      }
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
  // VALU-lean by construction: position math once per accumulator register (not per channel tile),
  // pooling windows are 1 or 2 (shifts, no integer division), dropout from a 32-bit mix hash.
  float s0[NT], s1[NT];
  float bias[NT], esc[NT], esh[NT];
  bool nok[NT];
#pragma unroll
  for (int j = 0; j < NT; ++j) {
    s0[j] = 0.f; s1[j] = 0.f;
    const int n = n0 + 32 * j + li;
    nok[j] = n < p.N;
    bias[j] = (p.bias && nok[j]) ? p.bias[n] : 0.f;
    esc[j] = 0.f; esh[j] = 0.f;
    if (EPI == EPI_GLU_POOL || EPI == EPI_GLU_BWD) {
      if (nok[j]) { esc[j] = p.e_scale[n]; esh[j] = p.e_shift[n]; }
    }
  }
  if (EPI == EPI_GLU_POOL) __syncthreads();  // As/Bs are recycled as the pooling stage
  float* Cs = smem;                          // [128][BN+1]
  const int sph = p.ph >> 1, spw = p.pw >> 1;  // pooling windows are 1 or 2
  const float inv_pool = 1.0f / (float)(p.ph * p.pw);
  const uint32_t dkey = drop_key(p.rng_stream, p.seed);
  const uint32_t dthr = drop_threshold(p.drop_p);
  const float dscale = p.drop_p > 0.f ? 1.0f / (1.0f - p.drop_p) : 1.0f;
  const int nbase = n0 + li;

  // Side inputs (pre-BN activation, pooled gradient, residual) are fetched for FOUR accumulator rows at a time
  // before any store: a store to `out` may alias those loads as far as the compiler knows (and does alias for the
  // in-place residual form), which would otherwise serialise one exposed global-load latency per element.
#pragma unroll
  for (int rg = 0; rg < 4; ++rg) {
    float ev[4][NT], dv[4][NT], rv[4][NT];
    size_t posv[4];
    int mmv[4];
    bool pokv[4];
#pragma unroll
    for (int rr = 0; rr < 4; ++rr) {
      const int r = rg * 4 + rr;
      const int mm = wave * 32 + crow(r, lh);
      const int gh = th0 + (mm >> P.lgTW), gw = tw0 + (mm & (p.TW - 1));
      const bool pok = gh < (p.valid_h > 0 ? p.valid_h : p.H) && gw < (p.valid_w > 0 ? p.valid_w : p.W);
      const size_t pos = ((size_t)nb * p.H + gh) * p.W + gw;
      mmv[rr] = mm; pokv[rr] = pok; posv[rr] = pos;
      bool pooled_ok = false;
      const float* dprow = nullptr;
      if (EPI == EPI_GLU_BWD) {
        const int gph = gh >> sph, gpw = gw >> spw;
        pooled_ok = pok && gph < p.Hp && gpw < p.Wp;
        dprow = p.e_dpool + (((size_t)nb * p.Hp + gph) * p.Wp + gpw) * p.N + nbase;
      }
#pragma unroll
      for (int j = 0; j < NT; ++j) {
        const bool ok = pok && nok[j];
        ev[rr][j] = 0.f; dv[rr][j] = 0.f; rv[rr][j] = 0.f;
        if (EPI == EPI_GLU_POOL || EPI == EPI_GLU_BWD || EPI == EPI_ADD_STATS2)
          if (ok) ev[rr][j] = p.e_src[pos * p.e_pitch + nbase + 32 * j];
        if (EPI == EPI_GLU_BWD)
          if (pooled_ok && nok[j]) dv[rr][j] = dprow[32 * j];
        if (EPI == EPI_ADD_STATS2)
          if (ok) rv[rr][j] = p.out2[pos * p.out_pitch + nbase + 32 * j];
      }
    }
#pragma unroll
    for (int rr = 0; rr < 4; ++rr) {
      const int r = rg * 4 + rr;
      const size_t pos = posv[rr];
      float* orow = p.out + pos * p.out_pitch + nbase;
      const uint64_t ebase = (uint64_t)pos * p.N + nbase;
#pragma unroll
      for (int j = 0; j < NT; ++j) {
        const bool ok = pokv[rr] && nok[j];
        const float v = acc[j][r] + bias[j];
        if (EPI == EPI_PLAIN) {
          if (ok) orow[32 * j] = v;
        } else if (EPI == EPI_STATS) {
          if (ok) {
            orow[32 * j] = v;
            s0[j] += v;
            s1[j] = fmaf(v, v, s1[j]);
          }
        } else if (EPI == EPI_GLU_POOL) {
          float res = 0.f;
          if (ok) {
            const float xn = fmaf(ev[rr][j], esc[j], esh[j]);
            res = v * sigmoid_fast(xn) * drop_mul(ebase + 32 * j, dkey, dthr, dscale);
          }
          Cs[mmv[rr] * (BN + 1) + 32 * j + li] = res;
        } else if (EPI == EPI_GLU_BWD) {
          if (ok) {
            const float xn = fmaf(ev[rr][j], esc[j], esh[j]);
            const float sg = sigmoid_fast(xn);
            const float dres = dv[rr][j] * inv_pool * drop_mul(ebase + 32 * j, dkey, dthr, dscale);
            const float dlin = dres * sg;
            orow[32 * j] = dlin;
            p.out2[pos * p.out_pitch + nbase + 32 * j] = dres * v * sg * (1.0f - sg);
            s0[j] += dlin;
          }
        } else {  // EPI_ADD_STATS2: g = acc + residual ; stats = (sum g, sum g*y)
          if (ok) {
            const float g = v + rv[rr][j];
            orow[32 * j] = g;
            s0[j] += g;
            s1[j] = fmaf(g, ev[rr][j], s1[j]);
          }
        }
      }
    }
  }

  if (EPI == EPI_GLU_POOL) {
    __syncthreads();
    const int lgtpw = P.lgTW - spw;
    const int tpw = 1 << lgtpw, tph = p.TH >> sph;
    for (int e = tid; e < tph * tpw * BN; e += IG_THREADS) {
      const int n = e % BN, pp = e / BN;
      const int pr = pp >> lgtpw, pc = pp & (tpw - 1);
      const int gph = (th0 >> sph) + pr, gpw = (tw0 >> spw) + pc;
      if (gph < p.Hp && gpw < p.Wp && n0 + n < p.N) {
        const float* c0 = Cs + (((pr << sph) << P.lgTW) + (pc << spw)) * (BN + 1) + n;
        float s = c0[0];
        if (spw) s += c0[BN + 1];
        if (sph) {
          s += c0[p.TW * (BN + 1)];
          if (spw) s += c0[(p.TW + 1) * (BN + 1)];
        }
        p.out[(((size_t)nb * p.Hp + gph) * p.Wp + gpw) * p.out_pitch + n0 + n] = s * inv_pool;
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
  int PW, PH, PP, lgTW, dy_off, ntiles, nct, ntw;  // nct = CC/32, ntw = 32-wide dy tiles per workgroup
  int CC, lgc4, pw_magic;                          // input-channel chunk per workgroup (grid.z), log2(CC/4)
  int pack2;                                       // CIN <= 16: two taps share one 32-row MFMA tile
};

// NW waves per workgroup share one staged (activation patch, dy tile) pair: 8 waves double the resident waves per
// byte of LDS, which is what hides the tile-load latency here (occupancy, not prefetching, is the lever).
template <int MAXS, int NW>
__global__ __launch_bounds__(NW * 64) void wgrad_kernel(const WgradParams P) {
  constexpr int NTHR = NW * 64;
  const BsedWgradDesc& p = P.d;
  extern __shared__ __align__(16) float smem[];
  const int XP = P.CC + 1;
  const int cz0 = blockIdx.z * P.CC;
  float* Xs = smem;
  float* DYs = smem + P.dy_off;  // [128][32*ntw]
  const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63, li = lane & 31, lh = lane >> 5;
  const int DYW = 32 * P.ntw;
  const int n0 = blockIdx.y * DYW;
  const int PW = P.PW;
  const int ntap_items = P.pack2 ? (p.ntaps + 1) / 2 : p.ntaps;
  const int nitems = ntap_items * P.nct * P.ntw;

  int xoff[MAXS], boff[MAXS];
  bool valid[MAXS];
  f32x16 acc[MAXS];
#pragma unroll
  for (int s = 0; s < MAXS; ++s) {
    const int it = wave + NW * s;
    valid[s] = it < nitems;
    // item -> (input-channel tile, dy tile, tap); a wave's items share the channel tile when ntw == 4
    const int cit = valid[s] ? it % P.nct : 0, rr = valid[s] ? it / P.nct : 0;
    const int nt = rr % P.ntw, tap = rr / P.ntw;
    if (P.pack2) {
      // rows 0..15 of the tile: tap 2*tap, rows 16..31: tap 2*tap+1 (channels 16..31 of the staged patch are zero,
      // which is what the upper rows read when the second tap does not exist)
      const int ta = 2 * tap, tb = 2 * tap + 1;
      if (li < 16 || tb >= p.ntaps) xoff[s] = (p.dh[ta] * PW + p.dw[ta]) * XP + li;
      else xoff[s] = (p.dh[tb] * PW + p.dw[tb]) * XP + (li - 16);
    } else {
      xoff[s] = (p.dh[tap] * PW + p.dw[tap]) * XP + cit * 32 + li;
    }
    boff[s] = nt * 32 + li;
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
    const int c4n = 1 << P.lgc4;
    const int x_total = P.PP * c4n;
    for (int e0 = tid; e0 < x_total; e0 += 4 * NTHR) {
      float4 v[4];
      bool okv[4];
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        const int e = e0 + u * NTHR;
        const int c4 = e & (c4n - 1), pos = e >> P.lgc4;
        const int pr = (pos * P.pw_magic) >> 20, pc = pos - pr * PW;
        const int gh = th0 - p.hh + pr, gw = tw0 - p.hw + pc;
        const int cg = cz0 + 4 * c4;
        v[u] = make_float4(0.f, 0.f, 0.f, 0.f);
        okv[u] = e < x_total && gh >= 0 && gh < p.H && gw >= 0 && gw < p.W && cg < p.CIN;
        if (okv[u]) v[u] = *reinterpret_cast<const float4*>(inb + ((size_t)gh * p.W + gw) * p.in_pitch + cg);
      }
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        const int e = e0 + u * NTHR;
        if (e < x_total) {
          const int c4 = e & (c4n - 1), pos = e >> P.lgc4;
          if (p.a_scale && okv[u]) {
            const int cg = cz0 + 4 * c4;
            const float4 sc = *reinterpret_cast<const float4*>(p.a_scale + cg);
            const float4 sh = *reinterpret_cast<const float4*>(p.a_shift + cg);
            v[u].x = fmaf(v[u].x, sc.x, sh.x); v[u].y = fmaf(v[u].y, sc.y, sh.y);
            v[u].z = fmaf(v[u].z, sc.z, sh.z); v[u].w = fmaf(v[u].w, sc.w, sh.w);
          }
          float* dst = Xs + pos * XP + 4 * c4;
          dst[0] = v[u].x; dst[1] = v[u].y; dst[2] = v[u].z; dst[3] = v[u].w;
        }
      }
    }
    const int n4n = 8 * P.ntw;
    const int lgn4 = P.ntw == 4 ? 5 : (P.ntw == 2 ? 4 : 3);
    for (int e0 = tid; e0 < IG_TILE_M * n4n; e0 += 4 * NTHR) {
      float4 v[4];
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        const int e = e0 + u * NTHR;
        const int n4 = e & (n4n - 1), mm = e >> lgn4;
        const int gh = th0 + (mm >> P.lgTW), gw = tw0 + (mm & (p.TW - 1));
        v[u] = make_float4(0.f, 0.f, 0.f, 0.f);
        if (e < IG_TILE_M * n4n && gh < p.H && gw < p.W && n0 + 4 * n4 < p.N)
          v[u] = *reinterpret_cast<const float4*>(dyb + ((size_t)gh * p.W + gw) * p.dy_pitch + n0 + 4 * n4);
      }
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        const int e = e0 + u * NTHR;
        const int n4 = e & (n4n - 1), mm = e >> lgn4;
        if (e < IG_TILE_M * n4n) *reinterpret_cast<float4*>(DYs + mm * DYW + 4 * n4) = v[u];
      }
    }
    __syncthreads();
#pragma unroll 2
    for (int kp = 0; kp < IG_TILE_M; kp += 2) {
      const int mk = kp + lh;
      const float* dyrow = DYs + mk * DYW;
      const float* xrow = Xs + (((mk >> P.lgTW) + p.hh) * PW + (mk & (p.TW - 1)) + p.hw) * XP;
      // straight-line on purpose: padded slots (it >= nitems) repeat item 0 and are dropped in the epilogue, so
      // the LDS reads of all slots issue ahead of the MFMAs instead of one exposed LDS latency per MFMA
      float av[MAXS], bv[MAXS];
#pragma unroll
      for (int s = 0; s < MAXS; ++s) {
        av[s] = xrow[xoff[s]];
        bv[s] = dyrow[boff[s]];
      }
#pragma unroll
      for (int s = 0; s < MAXS; ++s) acc[s] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[s], bv[s], acc[s], 0, 0, 0);
    }
  }
  const int NPo = gridDim.y * DYW;
#pragma unroll
  for (int s = 0; s < MAXS; ++s) {
    if (valid[s]) {
      const int it = wave + NW * s;
      const int cit = it % P.nct, rr = it / P.nct;
      const int nt = rr % P.ntw, tap = rr / P.ntw;
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        int ci = cz0 + cit * 32 + crow(r, lh), tp = tap;
        if (P.pack2) {
          const int row = crow(r, lh);
          tp = 2 * tap + (row >> 4);
          ci = row & 15;
          if (tp >= p.ntaps) continue;
        }
        p.part[(((size_t)blockIdx.x * p.ntaps + tp) * p.CINP + ci) * NPo + n0 + nt * 32 + li] = acc[s][r];
      }
    }
  }
}


// ---------------------------------------------------------------------------------------------
// weight gradient with split-fp32 operands on the bf16 matrix cores (bf16x3, see igemm3.hip).  K of the GEMM is the
// position index, so a lane's fragment is 8 POSITIONS of one channel:
//   * dy is staged TRANSPOSED ([n][128 positions], bf16 hi / lo planes) -> one 16-byte read per fragment, and a wave
//     whose work items all share the dy tile reads it once per K step;
//   * the activation patch stays position-major (as it arrives from HBM) in two bf16 planes (hi, lo); the tap-shifted
//     fragment comes out of gfx950's transposing LDS read: one ds_read_b64_tr_b16 hands every lane 4 consecutive
//     positions of its channel, whatever the row pitch, so a tap shift is an address offset.  4 such reads (2 LDS cycles
//     each, no VALU) replace the 8 ds_read_b32 + 8 v_perm of a word-packed patch, which had made this kernel
//     LDS-bandwidth bound (4 SIMDs x 24 LDS cycles per 96 MFMA cycles).
// Rows of 32 channels are 64-byte chunks; a fragment read touches 4 consecutive patch columns of one chunk, so the
// chunk index is XOR-swizzled with the patch COLUMN (invariant under row steps): 4 columns x 64 B tile the 64 banks.
// ---------------------------------------------------------------------------------------------
typedef short w3_bf16x8 __attribute__((ext_vector_type(8)));
typedef short w3_bf16x4 __attribute__((ext_vector_type(4)));
typedef __attribute__((address_space(3))) unsigned short w3_lds_u16;
typedef __attribute__((address_space(3))) w3_bf16x4 w3_lds_v4;
typedef uint32_t w3_u32x2 __attribute__((ext_vector_type(2)));
typedef float w3_f32x4 __attribute__((ext_vector_type(4)));
typedef __attribute__((address_space(3))) w3_u32x2 w3_lds_u2;

// physical 32-channel chunk of logical chunk `ch` in the row of position / patch column pc (nch = chunks per row: 1, 2
// or 4): 4 consecutive pc x 64 B then tile the 64 banks
__device__ __forceinline__ int w3_chunk(int ch, int pc, int nch) {
  return ch ^ ((nch == 2 ? (pc >> 1) : pc) & (nch - 1));
}

// 8 consecutive positions of this lane's channel: two transposing reads of 4 rows each.  a0 / a1 are LDS BYTE
// addresses (plane base included, so that no base add is left in the K loop)
__device__ __forceinline__ w3_bf16x8 w3_frag(uint32_t a0, uint32_t a1) {
  const w3_bf16x4 u = __builtin_amdgcn_ds_read_tr16_b64_v4i16((w3_lds_v4*)(uintptr_t)a0);
  const w3_bf16x4 v = __builtin_amdgcn_ds_read_tr16_b64_v4i16((w3_lds_v4*)(uintptr_t)a1);
  return __builtin_shufflevector(u, v, 0, 1, 2, 3, 4, 5, 6, 7);
}

// 4 channels of one position -> the two planes (ABF, the bf16 mode: the hi plane alone)
template <int ABF = 0>
__device__ __forceinline__ void w3_store4(w3_lds_u16* hi_plane, w3_lds_u16* lo_plane, int o, const float4& v) {
  w3_u32x2 hi, lo;
  uint32_t h0, l0, h1, l1;
  bsed_split2(v.x, v.y, h0, l0);
  bsed_split2(v.z, v.w, h1, l1);
  hi[0] = h0; hi[1] = h1; lo[0] = l0; lo[1] = l1;
  *(w3_lds_u2*)(hi_plane + o) = hi;
  if (!ABF) *(w3_lds_u2*)(lo_plane + o) = lo;
}
// activation loads / stores through a pointer computed in ELEMENTS from the tensor base (ABF: the tensor is bf16)
template <int ABF> __device__ __forceinline__ float4 w3_ld4(const float* base, const float* q) {
  const f32x4 t = act_ld4<ABF>(base, (size_t)(q - base));
  return make_float4(t[0], t[1], t[2], t[3]);
}
template <int ABF> __device__ __forceinline__ void w3_st4(float* base, float* q, const float4& v) {
  act_st4<ABF>(base, (size_t)(q - base), f32x4{v.x, v.y, v.z, v.w});
}
template <int ABF>
__device__ __forceinline__ f32x16 w3_mfma(const w3_bf16x8& a_hi, const w3_bf16x8& a_lo, const w3_bf16x8& b_hi,
                                          const w3_bf16x8& b_lo, f32x16 acc) {
  if (!ABF) {
    acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a_lo, b_hi, acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a_hi, b_lo, acc, 0, 0, 0);
  }
  return __builtin_amdgcn_mfma_f32_32x32x16_bf16(a_hi, b_hi, acc, 0, 0, 0);
}

#ifndef W3_SMALL_WPE
#define W3_SMALL_WPE 4
#endif
// the 1- and 2-slot 4-wave forms (conv1 16 -> 32 channels: 39 KB of LDS) are pinned to 128 registers so that FOUR
// workgroups share a CU: 0.753 (three, 136 registers) -> 0.651 ms, 4.9 TB/s (A/B: -DW3_SMALL_WPE=2)
#define W3_WPE(MAXS, NW) ((MAXS) <= 2 && (NW) == 4 ? W3_SMALL_WPE : 2)
template <int MAXS, int NW, bool BS, int ABF>
__global__ __launch_bounds__(NW * 64) __attribute__((amdgpu_waves_per_eu(W3_WPE(MAXS, NW)))) void wgrad3_kernel(const WgradParams P) {
  constexpr int NTHR = NW * 64;
  const BsedWgradDesc& p = P.d;
  extern __shared__ __align__(16) uint32_t smw[];
  const int CC = P.CC;
  const int cz0 = blockIdx.z * CC;
  const int DYW = 32 * P.ntw;
  w3_lds_u16* Xh = (w3_lds_u16*)smw;               // [PP][CC] bf16, chunk-swizzled
  w3_lds_u16* Xl = Xh + P.PP * CC;
  w3_lds_u16* DYh = Xh + 2 * P.dy_off;             // [128][DYW] bf16, chunk-swizzled
  w3_lds_u16* DYl = DYh + IG_TILE_M * DYW;
  const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63, li = lane & 31, lh = lane >> 5;
  const int n0 = blockIdx.y * DYW;
  const int PW = P.PW;
  const int ntap_items = P.pack2 ? (p.ntaps + 1) / 2 : p.ntaps;
  const int nitems = ntap_items * P.nct * P.ntw;
  // transposing read: lane 16g + 4q + c4 supplies the address of row (position) q, columns 4*c4 .. 4*c4+3 of its
  // 16-lane group's 4 x 16 block and receives column (lane & 15), rows 0..3 -- the operand layout of v_mfma_32x32x16
  // (row / column = lane & 31 = 16 * (g & 1) + block column, k = 8 * (lane >> 5) + 4 * read + block row)
  const int gq = (lane >> 2) & 3, gc = 4 * (lane & 3), gh = (lane >> 4) & 1;

  constexpr int NB_ = BS ? 1 : MAXS;
  const uint32_t xh0 = (uint32_t)(uintptr_t)Xh, dh0 = (uint32_t)(uintptr_t)DYh;
  const uint32_t xlo = 2 * P.PP * CC, dlo = 2 * IG_TILE_M * DYW;  // byte distance hi plane -> lo plane
  uint32_t xa[MAXS][2], xb[NB_][2];
  bool valid[MAXS];
  f32x16 acc[MAXS];
#pragma unroll
  for (int s = 0; s < MAXS; ++s) {
    const int it = wave + NW * s;
    valid[s] = it < nitems;
    const int cit = valid[s] ? it % P.nct : 0, rr = valid[s] ? it / P.nct : 0;
    const int nt = rr % P.ntw, tap = rr / P.ntw;
    int tp = tap, col = 16 * gh + gc;
    if (P.pack2) {
      // rows 0..15 of the tile: tap 2*tap, rows 16..31: tap 2*tap+1 (channels 16..31 of the staged patch are zero,
      // which is what the upper rows read when the second tap does not exist)
      const int tb = 2 * tap + 1;
      tp = 2 * tap;
      if (gh && tb < p.ntaps) { tp = tb; col = gc; }
    }
#pragma unroll
    for (int rd = 0; rd < 2; ++rd) {
      const int mk = 8 * lh + 4 * rd + gq;
      const int pr = (mk >> P.lgTW) + p.hh + p.dh[tp], pc = (mk & (p.TW - 1)) + p.hw + p.dw[tp];
      xa[s][rd] = xh0 + 2 * ((pr * PW + pc) * CC + (w3_chunk(cit, pc, P.nct) << 5) + col);
      if (!BS || s == 0) xb[BS ? 0 : s][rd] = dh0 + 2 * (mk * DYW + (w3_chunk(nt, mk, P.ntw) << 5) + 16 * gh + gc);
    }
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[s][r] = 0.f;
  }
  // BS: all of a wave's items sit on one dy tile (the wave stride NW is a multiple of channel tiles x dy tiles), so
  // its fragment is read once per K step
  const uint32_t kstep = 2 * (16 >> P.lgTW) * PW * CC;  // bytes; 16 positions = (16 / TW) tile rows (TW <= 16)
  const uint32_t bstep = 2 * 16 * DYW;
  // the fused input affine (BN apply of the previous layer): a thread's elements all sit in one channel quad, because
  // the thread count is a multiple of the quads per position -- loaded once, not once per element per tile
  float4 sc = make_float4(1.f, 1.f, 1.f, 1.f), sh = make_float4(0.f, 0.f, 0.f, 0.f);
  if (p.a_scale && cz0 + 4 * (tid & ((1 << P.lgc4) - 1)) < p.CIN) {
    sc = *reinterpret_cast<const float4*>(p.a_scale + cz0 + 4 * (tid & ((1 << P.lgc4) - 1)));
    sh = *reinterpret_cast<const float4*>(p.a_shift + cz0 + 4 * (tid & ((1 << P.lgc4) - 1)));
  }
  // BatchNorm backward on load (bn_y set): d_y = A g + B y + (C - B mean) for the thread's own channel quad of dy
  float4 cA = make_float4(1.f, 1.f, 1.f, 1.f), cB = make_float4(0.f, 0.f, 0.f, 0.f), cC = cB;
  if (p.bn_y && n0 + 4 * (tid & (8 * P.ntw - 1)) < p.N) {
    const int cn = n0 + 4 * (tid & (8 * P.ntw - 1));
    cA = *reinterpret_cast<const float4*>(p.bn_coef + cn);
    cB = *reinterpret_cast<const float4*>(p.bn_coef + p.N + cn);
    const float4 c2 = *reinterpret_cast<const float4*>(p.bn_coef + 2 * p.N + cn);
    const float4 mu = *reinterpret_cast<const float4*>(p.bn_mean + cn);
    cC = make_float4(fmaf(-cB.x, mu.x, c2.x), fmaf(-cB.y, mu.y, c2.y), fmaf(-cB.z, mu.z, c2.z), fmaf(-cB.w, mu.w, c2.w));
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
    // Both tiles arrive as coalesced float4 rows and stay position-major.  The loads of a round are all issued before
    // any is consumed: the staging phase is a chain of global-load latencies, so fewer, wider rounds is what counts
    // (a 180-position x 64-channel patch is 12 float4 per thread: two rounds of 6 instead of three of 4).
    constexpr int UX = 6, UD = 8;
    const int c4n = 1 << P.lgc4;
    const int x_total = P.PP * c4n;
    for (int e0 = tid; e0 < x_total; e0 += UX * NTHR) {
      float4 v[UX];
      bool okv[UX];
      int ov[UX];
#pragma unroll
      for (int u = 0; u < UX; ++u) {
        const int e = e0 + u * NTHR;
        const int c4 = e & (c4n - 1), pos = e >> P.lgc4;
        const int pr = (pos * P.pw_magic) >> 20, pc = pos - pr * PW;
        const int gh_ = th0 - p.hh + pr, gw = tw0 - p.hw + pc;
        const int cg = cz0 + 4 * c4;
        ov[u] = pos * CC + (w3_chunk(c4 >> 3, pc, P.nct) << 5) + 4 * (c4 & 7);
        v[u] = make_float4(0.f, 0.f, 0.f, 0.f);
        okv[u] = e < x_total && gh_ >= 0 && gh_ < p.H && gw >= 0 && gw < p.W && cg < p.CIN;
        if (okv[u]) v[u] = w3_ld4<ABF>(p.in, inb + ((size_t)gh_ * p.W + gw) * p.in_pitch + cg);
      }
#pragma unroll
      for (int u = 0; u < UX; ++u) {
        const int e = e0 + u * NTHR;
        if (e < x_total) {
          if (p.a_scale && okv[u]) {
            v[u].x = fmaf(v[u].x, sc.x, sh.x); v[u].y = fmaf(v[u].y, sc.y, sh.y);
            v[u].z = fmaf(v[u].z, sc.z, sh.z); v[u].w = fmaf(v[u].w, sc.w, sh.w);
          }
          w3_store4<ABF>(Xh, Xl, ov[u], v[u]);
        }
      }
    }
    const int n4n = 8 * P.ntw, lgn4 = P.ntw == 4 ? 5 : (P.ntw == 2 ? 4 : 3);
    const int d_total = IG_TILE_M * n4n;
    for (int e0 = tid; e0 < d_total; e0 += UD * NTHR) {
      float4 v[UD];
#pragma unroll
      for (int u = 0; u < UD; ++u) {
        const int e = e0 + u * NTHR;
        const int n4 = e & (n4n - 1), mm = e >> lgn4;
        const int gh_ = th0 + (mm >> P.lgTW), gw = tw0 + (mm & (p.TW - 1));
        v[u] = make_float4(0.f, 0.f, 0.f, 0.f);
        if (e < d_total && gh_ < p.H && gw < p.W && n0 + 4 * n4 < p.N) {
          const size_t o = ((size_t)nb * p.H * p.W + (size_t)gh_ * p.W + gw) * p.dy_pitch + n0 + 4 * n4;
          v[u] = w3_ld4<ABF>(p.dy, p.dy + o);
          if (p.bn_y) {
            // BatchNorm backward on load (NTHR is a multiple of n4n: the channel quad is the thread's own, cA/cB/cC)
            const float4 yv = w3_ld4<ABF>(p.bn_y, p.bn_y + o);
            v[u].x = fmaf(cA.x, v[u].x, fmaf(cB.x, yv.x, cC.x)); v[u].y = fmaf(cA.y, v[u].y, fmaf(cB.y, yv.y, cC.y));
            v[u].z = fmaf(cA.z, v[u].z, fmaf(cB.z, yv.z, cC.z)); v[u].w = fmaf(cA.w, v[u].w, fmaf(cB.w, yv.w, cC.w));
            if (p.dy_out && blockIdx.z == 0) w3_st4<ABF>(p.dy_out, p.dy_out + o, v[u]);
          }
        }
      }
#pragma unroll
      for (int u = 0; u < UD; ++u) {
        const int e = e0 + u * NTHR;
        const int n4 = e & (n4n - 1), mm = e >> lgn4;
        if (e < d_total) w3_store4<ABF>(DYh, DYl, mm * DYW + (w3_chunk(n4 >> 3, mm, P.ntw) << 5) + 4 * (n4 & 7), v[u]);
      }
    }
    __syncthreads();
    // slots go through the matrix cores in small groups: the fragments of group g+1 are requested before the MFMAs of
    // group g issue (sched_barrier pins that order), so one LDS latency is exposed per K step, not one per group,
    // and at most two groups of fragments are live (the 9-slot form has 144 accumulator registers)
    constexpr int SG = MAXS > 5 ? 2 : 3, NG = (MAXS + SG - 1) / SG;
    constexpr int SB = BS ? 1 : SG;
    uint32_t ko = 0, kb = 0;
#pragma unroll 1
    for (int kp = 0; kp < IG_TILE_M; kp += 16, ko += kstep, kb += bstep) {
      w3_bf16x8 ah[2][SG], al[2][SG], bh[2][SB], bl[2][SB];
#pragma unroll
      for (int g = 0; g <= NG; ++g) {
        if (g < NG) {
#pragma unroll
          for (int j = 0; j < SG; ++j) {
            const int sl = g * SG + j;
            if (sl < MAXS) {
              ah[g & 1][j] = w3_frag(xa[sl][0] + ko, xa[sl][1] + ko);
              al[g & 1][j] = w3_frag(xa[sl][0] + ko + xlo, xa[sl][1] + ko + xlo);
              if (!BS || sl == 0) {
                bh[g & 1][BS ? 0 : j] = w3_frag(xb[BS ? 0 : sl][0] + kb, xb[BS ? 0 : sl][1] + kb);
                bl[g & 1][BS ? 0 : j] = w3_frag(xb[BS ? 0 : sl][0] + kb + dlo, xb[BS ? 0 : sl][1] + kb + dlo);
              }
            }
          }
        }
        __builtin_amdgcn_sched_barrier(0);
        if (g > 0) {
#pragma unroll
          for (int j = 0; j < SG; ++j) {
            const int sl = (g - 1) * SG + j;
            if (sl < MAXS) {
              const w3_bf16x8 b_hi = BS ? bh[0][0] : bh[(g - 1) & 1][BS ? 0 : j];
              const w3_bf16x8 b_lo = BS ? bl[0][0] : bl[(g - 1) & 1][BS ? 0 : j];
              acc[sl] = w3_mfma<ABF>(ah[(g - 1) & 1][j], al[(g - 1) & 1][j], b_hi, b_lo, acc[sl]);
            }
          }
          __builtin_amdgcn_sched_barrier(0);
        }
      }
    }
  }
  const int NPo = gridDim.y * DYW;
#pragma unroll
  for (int s = 0; s < MAXS; ++s) {
    if (valid[s]) {
      const int it = wave + NW * s;
      const int cit = it % P.nct, rr = it / P.nct;
      const int nt = rr % P.ntw, tap = rr / P.ntw;
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        int ci = cz0 + cit * 32 + crow(r, lh), tp = tap;
        if (P.pack2) {
          const int row = crow(r, lh);
          tp = 2 * tap + (row >> 4);
          ci = row & 15;
          if (tp >= p.ntaps) continue;
        }
        p.part[(((size_t)blockIdx.x * p.ntaps + tp) * p.CINP + ci) * NPo + n0 + nt * 32 + li] = acc[s][r];
      }
    }
  }
}

// ---------------------------------------------------------------------------------------------
// 1-tap weight gradient (GLU linears, GRU input / recurrent weights) as a STREAMING kernel.
//
// dW (CIN x N) = sum over positions of in^T dy reads each activation byte once and does 2 N (or 2 CIN) FLOP per byte:
// a read-only, HBM-bound pass (a tuned read-only stream gets 6.3 TB/s on this part, tools/probe/hbm_read_probe.hip).
// wgrad3_kernel runs these shapes with ONE 128-position tile in 128 KB of LDS: a single workgroup per CU whose loads,
// operand split and MFMAs follow each other -- nothing is in flight while it computes (3.0-3.2 TB/s on the large shapes,
// 1.1-1.7 on the GRU ones).  Here a workgroup (8 waves, one per CU) owns a 128 x 128 slab of dW and streams 64-position
// tiles through TWO LDS stages (bf16 hi / lo planes, 64 KB each) with the global loads running TWO tiles ahead in
// registers: while tile t is in the matrix cores, tile t+1 waits in registers or is being split into the other stage
// and tile t+2 is in flight -- 64-128 KB of loads outstanding per CU at any time, one barrier per tile.
// Operand images, swizzle and the transposing fragment reads are those of wgrad3_kernel (results are bit-identical:
// same products, same accumulation order per slab; only the tile boundaries of the partial slabs differ).
// ---------------------------------------------------------------------------------------------
#define W1_TP 64
#define W1_THREADS 512
#define W1_PLANE (W1_TP * 128 * 2)   // bytes of one bf16 plane of a stage
#define W1_STAGE (4 * W1_PLANE)      // Xh | Xl | DYh | DYl

struct W1Set { float4 x[4], d[4]; };

template <bool SHIFT, int ABF>
__global__ __launch_bounds__(W1_THREADS) __attribute__((amdgpu_waves_per_eu(2, 2))) void wgrad1_kernel(const WgradParams P) {
  const BsedWgradDesc& p = P.d;
  extern __shared__ __align__(16) uint32_t smw[];
  const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63, li = lane & 31, lh = lane >> 5;
  const int cz0 = blockIdx.z * 128, n0 = blockIdx.y * 128;
  const long M = (long)p.NB * p.H * p.W;
  const int ntile = (int)((M + W1_TP - 1) / W1_TP);
  // this wave's two 32 x 32 blocks: input-channel chunk cit, dy tiles nt0 and nt0 + 2
  const int cit = wave & 3, nt0 = wave >> 2;
  const int gq = (lane >> 2) & 3, gc = 4 * (lane & 3), gh = (lane >> 4) & 1;
  const uint32_t base = (uint32_t)(uintptr_t)smw;
  uint32_t xa[2], xb[2][2];
#pragma unroll
  for (int rd = 0; rd < 2; ++rd) {
    const int mk = 8 * lh + 4 * rd + gq;
    xa[rd] = base + 2 * (mk * 128 + (w3_chunk(cit, mk, 4) << 5) + 16 * gh + gc);
#pragma unroll
    for (int s = 0; s < 2; ++s)
      xb[s][rd] = base + 2 * W1_PLANE + 2 * (mk * 128 + (w3_chunk(nt0 + 2 * s, mk, 4) << 5) + 16 * gh + gc);
  }
  f32x16 acc[2];
#pragma unroll
  for (int s = 0; s < 2; ++s)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[s][r] = 0.f;

  // loads: thread -> channel quad c4 (constant) of positions p0 + 16 u
  const int c4 = tid & 31, p0 = tid >> 5;
  const bool xok = cz0 + 4 * c4 < p.CIN, dok = n0 + 4 * c4 < p.N;
  const float* xsrc = p.in + cz0 + 4 * (xok ? c4 : 0);
  const float* dsrc = p.dy + n0 + 4 * (dok ? c4 : 0);
  float4 sc = make_float4(1.f, 1.f, 1.f, 1.f), sh = make_float4(0.f, 0.f, 0.f, 0.f);
  if (p.a_scale && xok) {
    sc = *reinterpret_cast<const float4*>(p.a_scale + cz0 + 4 * c4);
    sh = *reinterpret_cast<const float4*>(p.a_shift + cz0 + 4 * c4);
  }
  const int shift = SHIFT ? p.dh[0] * p.W + p.dw[0] : 0;
  const int so = p0 * 128 + (w3_chunk(c4 >> 3, p0, 4) << 5) + 4 * (c4 & 7);   // ushorts; + 16 u * 128 per u
  w3_lds_u16* lds = (w3_lds_u16*)smw;

  uint32_t xmask = 0, dmask = 0;   // validity bits of the set being loaded are recomputed at convert time
  auto issue = [&](int tile, W1Set& S) {
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      const long m = (long)tile * W1_TP + p0 + 16 * u;
      const long mc = m < M ? m : M - 1;   // clamped: no branch around the loads, idle rows are zeroed at convert time
      long ms = mc;
      if (SHIFT) {
        ms = mc + shift;
        ms = ms < 0 ? 0 : (ms >= M ? M - 1 : ms);
      }
      S.x[u] = w3_ld4<ABF>(p.in, xsrc + (size_t)ms * p.in_pitch);
      S.d[u] = w3_ld4<ABF>(p.dy, dsrc + (size_t)mc * p.dy_pitch);
    }
  };
  auto convert = [&](int tile, const W1Set& S, int stage) {
    w3_lds_u16* xh = lds + stage * (W1_STAGE / 2);
    w3_lds_u16* xl = xh + W1_PLANE / 2;
    w3_lds_u16* dh_ = xh + W1_PLANE;
    w3_lds_u16* dl = dh_ + W1_PLANE / 2;
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      const long m = (long)tile * W1_TP + p0 + 16 * u;
      bool okx = m < M && xok;
      if (SHIFT) {
        const int rem = (int)(m % ((long)p.H * p.W));
        const int h = rem / p.W + p.dh[0], w = rem % p.W + p.dw[0];
        okx = okx && h >= 0 && h < p.H && w >= 0 && w < p.W;
      }
      const bool okd = m < M && dok;
      float4 v = S.x[u];
      v.x = okx ? fmaf(v.x, sc.x, sh.x) : 0.f; v.y = okx ? fmaf(v.y, sc.y, sh.y) : 0.f;
      v.z = okx ? fmaf(v.z, sc.z, sh.z) : 0.f; v.w = okx ? fmaf(v.w, sc.w, sh.w) : 0.f;
      w3_store4<ABF>(xh, xl, so + 16 * u * 128, v);
      float4 g = S.d[u];
      g.x = okd ? g.x : 0.f; g.y = okd ? g.y : 0.f; g.z = okd ? g.z : 0.f; g.w = okd ? g.w : 0.f;
      w3_store4<ABF>(dh_, dl, so + 16 * u * 128, g);
    }
  };
  auto mma = [&](int stage) {
    const uint32_t so_ = stage * W1_STAGE;
#pragma unroll
    for (int k = 0; k < W1_TP / 16; ++k) {
      const uint32_t ko = so_ + k * 16 * 128 * 2;
      const w3_bf16x8 ah = w3_frag(xa[0] + ko, xa[1] + ko);
      const w3_bf16x8 al = w3_frag(xa[0] + ko + W1_PLANE, xa[1] + ko + W1_PLANE);
      w3_bf16x8 bh[2], bl[2];
#pragma unroll
      for (int s = 0; s < 2; ++s) {
        bh[s] = w3_frag(xb[s][0] + ko, xb[s][1] + ko);
        bl[s] = w3_frag(xb[s][0] + ko + W1_PLANE, xb[s][1] + ko + W1_PLANE);
      }
#pragma unroll
      for (int s = 0; s < 2; ++s) {
        acc[s] = w3_mfma<ABF>(ah, al, bh[s], bl[s], acc[s]);
      }
    }
  };
  (void)xmask; (void)dmask;

  // tiles of this workgroup: t0, t0 + G, ...   (n of them; tile indices past the end load clamped rows that convert
  // zeroes and nobody multiplies)
  const int t0 = blockIdx.x, G = gridDim.x;
  const int n = t0 < ntile ? (ntile - t0 + G - 1) / G : 0;
  W1Set A, B;
  if (n > 0) {
    issue(t0, A);
    issue(t0 + G, B);
    convert(t0, A, 0);
    __syncthreads();
    for (int i = 0; i < n; i += 2) {
      // even step: tile i in stage 0; A is free, B holds tile i + 1
      issue(t0 + (i + 2) * G, A);
      mma(0);
      convert(t0 + (i + 1) * G, B, 1);
      __syncthreads();
      if (i + 1 >= n) break;
      // odd step: tile i + 1 in stage 1; B is free, A holds tile i + 2
      issue(t0 + (i + 3) * G, B);
      mma(1);
      convert(t0 + (i + 2) * G, A, 0);
      __syncthreads();
    }
  }
  const int NPo = gridDim.y * 128;
#pragma unroll
  for (int s = 0; s < 2; ++s)
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int ci = cz0 + cit * 32 + crow(r, lh);
      p.part[((size_t)blockIdx.x * p.CINP + ci) * NPo + n0 + (nt0 + 2 * s) * 32 + li] = acc[s][r];
    }
}

// ---------------------------------------------------------------------------------------------
// Producer / consumer form of the bf16x3 weight gradient for the multi-tap convolutions: ONE 8-wave workgroup per CU
// with two LDS tile buffers.  Waves 0-3 (one per SIMD) only run the MFMA loop of tile t; waves 4-7 (one per SIMD) only
// load, split and write tile t+1 into the other buffer; one barrier per tile hands the buffers over.  In
// wgrad3_kernel the two co-resident workgroups run in phase (stage, then compute), so the staging time adds to the
// MFMA time; here a SIMD always has one wave of each kind and the staging hides behind the matrix cores.
// Same tiles, same products, same accumulation order as wgrad3_kernel.
// ---------------------------------------------------------------------------------------------
#define W3P_UX 13  // float4 of the activation patch per producer thread (256 threads): PP * CC <= 13312 elements
#define W3P_UD 8   // float4 of the dy tile per producer thread: 128 positions x 64 channels

// GEO > 0: the tile geometry of one of the network's layer shapes is a compile-time constant: every LDS offset of the
// MFMA waves' K loop becomes an instruction immediate and the loop is unrolled (20 instead of 640 v_add per tile and no
// spills: -7 %).  GEO = 0 keeps everything at run time (any other shape).
struct W3Geo { int CC, lgTW, PW, PP, ntw; };
__host__ __device__ constexpr W3Geo w3_geo(int g) {
  return g == 1 ? W3Geo{64, 4, 18, 180, 2}     // 64 -> 128 channels on the 216 x 16 map
       : g == 2 ? W3Geo{64, 3, 10, 180, 2}     // 128 -> 128, 216 x 8
       : g == 3 ? W3Geo{32, 4, 18, 180, 2}     // 32 -> 64, 216 x 32
       : g == 4 ? W3Geo{64, 2, 6, 204, 1}      // 128 -> 128, 216 x 4
       : g == 5 ? W3Geo{32, 1, 4, 264, 2}      // 128 -> 128, 216 x 2
       : g == 6 ? W3Geo{32, 4, 18, 180, 1}     // 16 -> 32 (two taps per MFMA tile; also 32 -> 32), 432 x 64
                : W3Geo{0, 0, 0, 0, 0};
}
#define W3_NGEO 6

// Diagnostic build (-DW3P_STAMP, tools/build_variant.sh + tools/wgrad_stamp.py): s_memtime stamps of the fourth tile period
// of every workgroup -- slots 0..3 producer wave 4 (period start, loads landed, conversions + refills issued, barrier
// passed), 4..6 consumer wave 0 (period start, MFMAs issued, barrier passed), 7 HW_ID.
#ifdef W3P_STAMP
__device__ unsigned long long w3p_stamp_buf[1024 * 8];
extern "C" int bsed_w3p_stamps(unsigned long long* host) {
  return hipMemcpyFromSymbol(host, HIP_SYMBOL(w3p_stamp_buf), sizeof(w3p_stamp_buf)) == hipSuccess ? 0 : 1;
}
#define W3P_ST(slot, cond)                                                                                         \
  if (cond) {                                                                                                      \
    const unsigned long long t_ = __builtin_amdgcn_s_memtime();                                                    \
    if ((threadIdx.x & 63) == 0)                                                                                   \
      w3p_stamp_buf[((blockIdx.z * gridDim.y + blockIdx.y) * gridDim.x + blockIdx.x) % 1024 * 8 + (slot)] = t_;    \
  }
#else
#define W3P_ST(slot, cond)
#endif

template <int MAXS, int GEO, int ABF>
__global__ __launch_bounds__(512) __attribute__((amdgpu_waves_per_eu(2, 2))) void wgrad3p_kernel(const WgradParams P) {
  constexpr int NTHR = 256, NW = 4, UX = W3P_UX, UD = W3P_UD;
  const BsedWgradDesc& p = P.d;
  extern __shared__ __align__(16) uint32_t smw[];
  const int CC = P.CC;
  const int cz0 = blockIdx.z * CC;
  const int DYW = 32 * P.ntw;
  w3_lds_u16* Xh = (w3_lds_u16*)smw;
  w3_lds_u16* Xl = Xh + P.PP * CC;
  w3_lds_u16* DYh = Xh + 2 * P.dy_off;
  w3_lds_u16* DYl = DYh + IG_TILE_M * DYW;
  const int buf_u16 = 2 * P.dy_off + 2 * IG_TILE_M * DYW;  // ushorts per tile buffer
  const int n0 = blockIdx.y * DYW;
  const int PW = P.PW;

  if (threadIdx.x >= NTHR) {
    // ------------------------------------------------------------------ producer waves
    const int tid = threadIdx.x - NTHR;
    const int c4n = 1 << P.lgc4;
    const int x_total = P.PP * c4n;
    const int n4n = 8 * P.ntw, lgn4 = P.ntw == 4 ? 5 : (P.ntw == 2 ? 4 : 3);
    const int d_total = IG_TILE_M * n4n;
    // the fused input affine (BN apply of the previous layer): a thread's elements all sit in one channel quad
    float4 sc = make_float4(1.f, 1.f, 1.f, 1.f), sh = make_float4(0.f, 0.f, 0.f, 0.f);
    if (p.a_scale && cz0 + 4 * (tid & (c4n - 1)) < p.CIN) {
      sc = *reinterpret_cast<const float4*>(p.a_scale + cz0 + 4 * (tid & (c4n - 1)));
      sh = *reinterpret_cast<const float4*>(p.a_shift + cz0 + 4 * (tid & (c4n - 1)));
    }
    // BatchNorm backward on load: d_y = A g + B y + (C - B mean); a thread's dy elements all sit in one channel quad
    const bool bnb = p.bn_y != nullptr;
    const bool dy_store = bnb && p.dy_out != nullptr && blockIdx.z == 0;   // one CIN chunk writes d_y out
    float4 cA = make_float4(1.f, 1.f, 1.f, 1.f), cB = make_float4(0.f, 0.f, 0.f, 0.f), cC = cB;
    if (bnb) {
      // (a thread whose channel quad lies beyond N works on quad 0 instead: its loads stay inside the tensor, its LDS
      //  image is zeroed, and what it stores to dy_out is the value quad 0's owner stores there as well)
      const int cn = n0 + 4 * (tid & (n4n - 1)) < p.N ? n0 + 4 * (tid & (n4n - 1)) : 0;
      cA = *reinterpret_cast<const float4*>(p.bn_coef + cn);
      cB = *reinterpret_cast<const float4*>(p.bn_coef + p.N + cn);
      const float4 c2 = *reinterpret_cast<const float4*>(p.bn_coef + 2 * p.N + cn);
      const float4 mu = *reinterpret_cast<const float4*>(p.bn_mean + cn);
      cC = make_float4(fmaf(-cB.x, mu.x, c2.x), fmaf(-cB.y, mu.y, c2.y), fmaf(-cB.z, mu.z, c2.z), fmaf(-cB.w, mu.w, c2.w));
    }
    if constexpr (ABF) {
      // ---- bf16 activations: 16-byte pieces of EIGHT channels (half the load / store / LDS-write instructions of the
      // four-channel pieces), no hi / lo split: a piece goes to the hi plane as it arrives
      constexpr int UX8 = (UX + 1) / 2, UD8 = UD / 2;
      const int lgc8 = P.lgc4 - 1, c8n = 1 << lgc8, x_total8 = P.PP * c8n;
      const int n8n = 4 * P.ntw, lgn8 = lgn4 - 1, d_total8 = IG_TILE_M * n8n;
      // (same software pipeline and straight-line period as the fp32 producers below -- see the comment there)
      constexpr W3Geo Gm = w3_geo(GEO);
      const int x_tot = GEO ? Gm.PP * (Gm.CC / 8) : x_total8, d_tot = GEO ? IG_TILE_M * 4 * Gm.ntw : d_total8;
      const int gW = p.W, gH = p.H;
      const bool xch_ok = cz0 + 8 * (tid & (c8n - 1)) < p.CIN, dch_ok = n0 + 8 * (tid & (n8n - 1)) < p.N;
      const int xcoff = xch_ok ? cz0 + 8 * (tid & (c8n - 1)) : 0, dcoff = dch_ok ? n0 + 8 * (tid & (n8n - 1)) : 0;
      float cA8[8], cB8[8], cC8[8];
#pragma unroll
      for (int q = 0; q < 8; ++q) { cA8[q] = 1.f; cB8[q] = 0.f; cC8[q] = 0.f; }
      if (bnb) {
#pragma unroll
        for (int q = 0; q < 8; ++q) {
          cA8[q] = p.bn_coef[dcoff + q];
          cB8[q] = p.bn_coef[p.N + dcoff + q];
          cC8[q] = fmaf(-cB8[q], p.bn_mean[dcoff + q], p.bn_coef[2 * p.N + dcoff + q]);
        }
      }
      float sc8[8], sh8[8];
      const bool affine = p.a_scale != nullptr;
#pragma unroll
      for (int q = 0; q < 8; ++q) { sc8[q] = 1.f; sh8[q] = 0.f; }
      if (affine && xch_ok) {
#pragma unroll
        for (int q = 0; q < 8; ++q) {
          sc8[q] = p.a_scale[xcoff + q];
          sh8[q] = p.a_shift[xcoff + q];
        }
      }
      int xo[UX8], xrc[UX8], dyo[UD8], drc[UD8];
#pragma unroll
      for (int u = 0; u < UX8; ++u) {
        const int e = tid + u * NTHR;
        const int c8 = e & (c8n - 1), pos = e >> lgc8;
        const int pr = GEO ? pos / (GEO ? Gm.PW : 1) : (pos * P.pw_magic) >> 20, pc = pos - pr * PW;
        xo[u] = pos * CC + (w3_chunk(c8 >> 2, pc, P.nct) << 5) + 8 * (c8 & 3);
        xrc[u] = (e < x_tot && xch_ok) ? ((pr - p.hh) << 16) | ((pc - p.hw) & 0xffff) : (0x4000 << 16);
      }
#pragma unroll
      for (int u = 0; u < UD8; ++u) {
        const int e = tid + u * NTHR;
        const int n8 = e & (n8n - 1), mm = e >> lgn8;
        const int r = mm >> P.lgTW, c = mm & (p.TW - 1);
        dyo[u] = mm * DYW + (w3_chunk(n8 >> 2, mm, P.ntw) << 5) + 8 * (n8 & 3);
        drc[u] = (e < d_tot && dch_ok) ? (r << 16) | c : (0x4000 << 16);
      }
      const unsigned short* in16 = reinterpret_cast<const unsigned short*>(p.in);
      const unsigned short* dy16 = reinterpret_cast<const unsigned short*>(p.dy);
      const unsigned short* y16 = reinterpret_cast<const unsigned short*>(p.bn_y);
      unsigned short* out16 = reinterpret_cast<unsigned short*>(p.dy_out);
      auto unpack8 = [](const bsed_u32x4& r, float* v) {
#pragma unroll
        for (int q = 0; q < 4; ++q) { v[2 * q] = __uint_as_float(r[q] << 16); v[2 * q + 1] = __uint_as_float(r[q] & 0xFFFF0000u); }
      };
      auto pack8 = [](const float* v) {
        bsed_u32x4 r;
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          const f32x2 t = {v[2 * q], v[2 * q + 1]};
          r[q] = __builtin_bit_cast(uint32_t, __builtin_convertvector(t, bsed_bf16x2));
        }
        return r;
      };
      struct TGeo { int th0, tw0; size_t in_img, dy_img; };
      auto tgeo = [&](int tile) {
        int t = tile;
        const int tw_i = t % p.tilesW; t /= p.tilesW;
        const int th_i = t % p.tilesH;
        const int nb = t / p.tilesH;
        TGeo g;
        g.th0 = th_i * p.TH; g.tw0 = tw_i * p.TW;
        g.in_img = (size_t)nb * p.H * p.W * p.in_pitch;
        g.dy_img = (size_t)nb * p.H * p.W * p.dy_pitch;
        return g;
      };
      auto x_at = [&](int u, const TGeo& g, bool& ok) {
        const int gh_ = g.th0 + (xrc[u] >> 16), gw = g.tw0 + (int)(short)(xrc[u] & 0xffff);
        ok = (unsigned)gh_ < (unsigned)gH && (unsigned)gw < (unsigned)gW;
        return (uint32_t)((min(max(gh_, 0), gH - 1) * gW + min(max(gw, 0), gW - 1)) * p.in_pitch + xcoff) << 1;
      };
      auto d_at = [&](int u, const TGeo& g, bool& ok) {
        const int gh_ = g.th0 + (drc[u] >> 16), gw = g.tw0 + (drc[u] & 0xffff);
        ok = gh_ < gH && gw < gW;
        return (uint32_t)((min(gh_, gH - 1) * gW + min(gw, gW - 1)) * p.dy_pitch + dcoff) << 1;
      };
      auto ld16 = [](const unsigned short* base, size_t img, uint32_t byte_off) {
        return *reinterpret_cast<const bsed_u32x4*>(reinterpret_cast<const char*>(base + img) + byte_off);
      };
      const bsed_u32x4 zero = {0u, 0u, 0u, 0u};
      auto produce = [&](auto bnb_c, auto store_c, auto aff_c) {
        constexpr bool BNB = decltype(bnb_c)::value, STORE = decltype(store_c)::value, AFF = decltype(aff_c)::value;
        bsed_u32x4 vx[UX8], vd[UD8], vy[BNB ? UD8 : 1];
        int tile = blockIdx.x, buf = 0;
        if (tile >= P.ntiles) return;
        TGeo gc = tgeo(tile);
#pragma unroll
        for (int u = 0; u < UD8; ++u) {
          if (GEO && u * NTHR >= d_tot) continue;
          bool ok;
          const uint32_t o = d_at(u, gc, ok);
          vd[u] = ld16(dy16, gc.dy_img, o);
          if (BNB) vy[u] = ld16(y16, gc.dy_img, o);
        }
#pragma unroll
        for (int u = 0; u < UX8; ++u) {
          if (GEO && u * NTHR >= x_tot) continue;
          bool ok;
          vx[u] = ld16(in16, gc.in_img, x_at(u, gc, ok));
        }
        for (; tile < P.ntiles; buf ^= 1) {
          const int nxt = tile + gridDim.x;
          const TGeo gn = tgeo(nxt < P.ntiles ? nxt : tile);
          const int bo = buf * buf_u16;
#pragma unroll
          for (int u = 0; u < UD8; ++u) {
            if (GEO && u * NTHR >= d_tot) continue;
            bool ok, okn;
            const uint32_t o = d_at(u, gc, ok);
            bsed_u32x4 v = vd[u];
            if (BNB) {
              float g8[8], y8[8];
              unpack8(v, g8); unpack8(vy[u], y8);
#pragma unroll
              for (int q = 0; q < 8; ++q) g8[q] = fmaf(cA8[q], g8[q], fmaf(cB8[q], y8[q], cC8[q]));
              v = pack8(g8);
              // unconditional (see the fp32 producers): a clamped piece stores the value its position's owner stores
              if (STORE) *reinterpret_cast<bsed_u32x4*>(reinterpret_cast<char*>(out16 + gc.dy_img) + o) = v;
            }
            if (!ok) v = zero;
            if (tid + u * NTHR < d_tot)
              *reinterpret_cast<__attribute__((address_space(3))) bsed_u32x4*>(DYh + bo + dyo[u]) = v;
            const uint32_t on = d_at(u, gn, okn);
            vd[u] = ld16(dy16, gn.dy_img, on);
            if (BNB) vy[u] = ld16(y16, gn.dy_img, on);
          }
#pragma unroll
          for (int u = 0; u < UX8; ++u) {
            if (GEO && u * NTHR >= x_tot) continue;
            bool ok, okn;
            (void)x_at(u, gc, ok);
            bsed_u32x4 v = vx[u];
            if (AFF) {
              float x8[8];
              unpack8(v, x8);
#pragma unroll
              for (int q = 0; q < 8; ++q) x8[q] = fmaf(x8[q], sc8[q], sh8[q]);
              v = pack8(x8);
            }
            if (!ok) v = zero;
            if (tid + u * NTHR < x_tot)
              *reinterpret_cast<__attribute__((address_space(3))) bsed_u32x4*>(Xh + bo + xo[u]) = v;
            vx[u] = ld16(in16, gn.in_img, x_at(u, gn, okn));
          }
          gc = gn; tile = nxt;
          __syncthreads();
        }
      };
      using T_ = std::true_type; using F_ = std::false_type;
      if (!bnb) { if (affine) produce(F_{}, F_{}, T_{}); else produce(F_{}, F_{}, F_{}); }
      else if (!dy_store) { if (affine) produce(T_{}, F_{}, T_{}); else produce(T_{}, F_{}, F_{}); }
      else { if (affine) produce(T_{}, T_{}, T_{}); else produce(T_{}, T_{}, F_{}); }
      __syncthreads();  // pairs with the consumers' barrier after their last tile
      return;
    }
    // ---- fp32 activations.  Tile-invariant element geometry: LDS offset and patch row / col of each piece.  A thread's
    // pieces all sit in one channel quad (256 threads are a multiple of the quads per position).
    //
    // SOFTWARE PIPELINE over tiles: the registers of a piece are refilled with the NEXT tile's piece right after the
    // piece has been split into LDS, so a tile's loads have a whole tile period (one MFMA pass of the consumers) to
    // arrive.  (Loading, waiting and converting inside one period took longer than the consumers' MFMA pass: the matrix
    // cores idled behind the producers.)  Everything between two barriers is STRAIGHT-LINE code -- rows / columns outside
    // the map are clamped for the address and zeroed at conversion, the last period re-requests its own tile -- because
    // hipcc's wait-count insertion falls back to vmcnt(0) as soon as a load sits inside a branch, which would serialise
    // the refills against the conversions.
    constexpr W3Geo Gm = w3_geo(GEO);
    const int x_tot = GEO ? Gm.PP * (Gm.CC / 4) : x_total, d_tot = GEO ? IG_TILE_M * 8 * Gm.ntw : d_total;
    const int gW = p.W, gH = p.H;
    int xo[UX], xrc[UX], dyo[UD], drc[UD];
    const bool xch_ok = cz0 + 4 * (tid & (c4n - 1)) < p.CIN, dch_ok = n0 + 4 * (tid & (n4n - 1)) < p.N;
    const int xcoff = xch_ok ? cz0 + 4 * (tid & (c4n - 1)) : 0, dcoff = dch_ok ? n0 + 4 * (tid & (n4n - 1)) : 0;
#pragma unroll
    for (int u = 0; u < UX; ++u) {
      const int e = tid + u * NTHR;
      const int c4 = e & (c4n - 1), pos = e >> P.lgc4;
      const int pr = GEO ? pos / (GEO ? Gm.PW : 1) : (pos * P.pw_magic) >> 20, pc = pos - pr * PW;
      xo[u] = pos * CC + (w3_chunk(c4 >> 3, pc, P.nct) << 5) + 4 * (c4 & 7);
      xrc[u] = (e < x_tot && xch_ok) ? ((pr - p.hh) << 16) | ((pc - p.hw) & 0xffff) : (0x4000 << 16);
    }
#pragma unroll
    for (int u = 0; u < UD; ++u) {
      const int e = tid + u * NTHR;
      const int n4 = e & (n4n - 1), mm = e >> lgn4;
      const int r = mm >> P.lgTW, c = mm & (p.TW - 1);
      dyo[u] = mm * DYW + (w3_chunk(n4 >> 3, mm, P.ntw) << 5) + 4 * (n4 & 7);
      drc[u] = (e < d_tot && dch_ok) ? (r << 16) | c : (0x4000 << 16);
    }
    struct TGeo { int th0, tw0; size_t in_img, dy_img; };
    auto tgeo = [&](int tile) {
      int t = tile;
      const int tw_i = t % p.tilesW; t /= p.tilesW;
      const int th_i = t % p.tilesH;
      const int nb = t / p.tilesH;
      TGeo g;
      g.th0 = th_i * p.TH; g.tw0 = tw_i * p.TW;
      g.in_img = (size_t)nb * p.H * p.W * p.in_pitch;
      g.dy_img = (size_t)nb * p.H * p.W * p.dy_pitch;
      return g;
    };
    // byte offset of a piece inside its image (clamped into the map) and whether the piece is inside the map
    auto x_at = [&](int u, const TGeo& g, bool& ok) {
      const int gh_ = g.th0 + (xrc[u] >> 16), gw = g.tw0 + (int)(short)(xrc[u] & 0xffff);
      ok = (unsigned)gh_ < (unsigned)gH && (unsigned)gw < (unsigned)gW;
      return (uint32_t)((min(max(gh_, 0), gH - 1) * gW + min(max(gw, 0), gW - 1)) * p.in_pitch + xcoff) << 2;
    };
    auto d_at = [&](int u, const TGeo& g, bool& ok) {
      const int gh_ = g.th0 + (drc[u] >> 16), gw = g.tw0 + (drc[u] & 0xffff);
      ok = gh_ < gH && gw < gW;
      return (uint32_t)((min(gh_, gH - 1) * gW + min(gw, gW - 1)) * p.dy_pitch + dcoff) << 2;
    };
    auto ld16 = [](const float* base, size_t img, uint32_t byte_off) {
      return *reinterpret_cast<const w3_f32x4*>(reinterpret_cast<const char*>(base + img) + byte_off);
    };
    auto produce = [&](auto bnb_c, auto store_c) {
      constexpr bool BNB = decltype(bnb_c)::value, STORE = decltype(store_c)::value;
      w3_f32x4 vx[UX], vd[UD], vy[BNB ? UD : 1];  // native vectors: arrays of HIP float4 structs end up in scratch
      int tile = blockIdx.x, buf = 0;
      if (tile >= P.ntiles) return;
      TGeo gc = tgeo(tile);
#pragma unroll
      for (int u = 0; u < UD; ++u) {
        if (GEO && u * NTHR >= d_tot) continue;
        bool ok;
        const uint32_t o = d_at(u, gc, ok);
        vd[u] = ld16(p.dy, gc.dy_img, o);
        if (BNB) vy[u] = ld16(p.bn_y, gc.dy_img, o);
      }
#pragma unroll
      for (int u = 0; u < UX; ++u) {
        if (GEO && u * NTHR >= x_tot) continue;
        bool ok;
        vx[u] = ld16(p.in, gc.in_img, x_at(u, gc, ok));
      }
      for (; tile < P.ntiles; buf ^= 1) {
        const int nxt = tile + gridDim.x;
        const TGeo gn = tgeo(nxt < P.ntiles ? nxt : tile);
        // (the buffer being written was last read before the previous barrier)
        const int bo = buf * buf_u16;
#ifdef W3P_STAMP
        const bool st_ = tile == (int)blockIdx.x + 3 * (int)gridDim.x && tid < 64;
        W3P_ST(0, st_);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        W3P_ST(1, st_);
#endif
#pragma unroll
        for (int u = 0; u < UD; ++u) {
          if (GEO && u * NTHR >= d_tot) continue;
          bool ok, okn;
          const uint32_t o = d_at(u, gc, ok);
          w3_f32x4 v = vd[u];
          if (BNB) {
            // BatchNorm backward on load: d_y is formed here and (one CIN chunk only) written out for the dgrad
            v[0] = fmaf(cA.x, v[0], fmaf(cB.x, vy[u][0], cC.x));
            v[1] = fmaf(cA.y, v[1], fmaf(cB.y, vy[u][1], cC.y));
            v[2] = fmaf(cA.z, v[2], fmaf(cB.z, vy[u][2], cC.z));
            v[3] = fmaf(cA.w, v[3], fmaf(cB.w, vy[u][3], cC.w));
            // unconditional: a piece outside the map was loaded from the clamped position, so it carries -- and stores
            // -- exactly the value the owner of that position stores (no branch around a memory instruction)
            if (STORE) *reinterpret_cast<w3_f32x4*>(reinterpret_cast<char*>(p.dy_out + gc.dy_img) + o) = v;
          }
          const float4 w = make_float4(ok ? v[0] : 0.f, ok ? v[1] : 0.f, ok ? v[2] : 0.f, ok ? v[3] : 0.f);
          if (tid + u * NTHR < d_tot) w3_store4<ABF>(DYh, DYl, bo + dyo[u], w);
          const uint32_t on = d_at(u, gn, okn);
          vd[u] = ld16(p.dy, gn.dy_img, on);
          if (BNB) vy[u] = ld16(p.bn_y, gn.dy_img, on);
        }
#pragma unroll
        for (int u = 0; u < UX; ++u) {
          if (GEO && u * NTHR >= x_tot) continue;
          bool ok, okn;
          (void)x_at(u, gc, ok);
          const w3_f32x4 v = vx[u];
          const float4 w = make_float4(ok ? fmaf(v[0], sc.x, sh.x) : 0.f, ok ? fmaf(v[1], sc.y, sh.y) : 0.f,
                                       ok ? fmaf(v[2], sc.z, sh.z) : 0.f, ok ? fmaf(v[3], sc.w, sh.w) : 0.f);
          if (tid + u * NTHR < x_tot) w3_store4<ABF>(Xh, Xl, bo + xo[u], w);
          vx[u] = ld16(p.in, gn.in_img, x_at(u, gn, okn));
        }
#ifdef W3P_STAMP
        W3P_ST(2, st_);
#endif
        gc = gn; tile = nxt;
        __syncthreads();
#ifdef W3P_STAMP
        W3P_ST(3, st_);
#endif
      }
    };
    if (!bnb) produce(std::false_type{}, std::false_type{});
    else if (!dy_store) produce(std::true_type{}, std::false_type{});
    else produce(std::true_type{}, std::true_type{});
    __syncthreads();  // pairs with the consumers' barrier after their last tile
    return;
  }

  // -------------------------------------------------------------------- consumer waves
  const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63, li = lane & 31, lh = lane >> 5;
  const int ntap_items = P.pack2 ? (p.ntaps + 1) / 2 : p.ntaps;
  const int nitems = ntap_items * P.nct * P.ntw;
  const int gq = (lane >> 2) & 3, gc = 4 * (lane & 3), gh = (lane >> 4) & 1;  // see wgrad3_kernel
  const uint32_t xh0 = (uint32_t)(uintptr_t)Xh, dh0 = (uint32_t)(uintptr_t)DYh;
  const uint32_t xlo = 2 * P.PP * CC, dlo = 2 * IG_TILE_M * DYW;
  uint32_t xa[MAXS][2], xb[2];
  bool valid[MAXS];
  f32x16 acc[MAXS];
#pragma unroll
  for (int s = 0; s < MAXS; ++s) {
    const int it = wave + NW * s;
    valid[s] = it < nitems;
    const int cit = valid[s] ? it % P.nct : 0, rr = valid[s] ? it / P.nct : 0;
    const int nt = rr % P.ntw, tap = rr / P.ntw;
    int tp = tap, col = 16 * gh + gc;
    if (P.pack2) {
      const int tb = 2 * tap + 1;
      tp = 2 * tap;
      if (gh && tb < p.ntaps) { tp = tb; col = gc; }
    }
#pragma unroll
    for (int rd = 0; rd < 2; ++rd) {
      const int mk = 8 * lh + 4 * rd + gq;
      const int pr = (mk >> P.lgTW) + p.hh + p.dh[tp], pc = (mk & (p.TW - 1)) + p.hw + p.dw[tp];
      xa[s][rd] = xh0 + 2 * ((pr * PW + pc) * CC + (w3_chunk(cit, pc, P.nct) << 5) + col);
      // the host only picks this kernel when all items of a wave share the dy tile (4 % (nct * ntw) == 0)
      if (s == 0) xb[rd] = dh0 + 2 * (mk * DYW + (w3_chunk(nt, mk, P.ntw) << 5) + 16 * gh + gc);
    }
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[s][r] = 0.f;
  }
  __syncthreads();  // tile 0 is in buffer 0
  int buf = 0;
  // fragments of slot group g+1 are requested before the MFMAs of group g issue (two groups live at a time)
  constexpr int SG = 2, NG = (MAXS + SG - 1) / SG;
  for (int tile = blockIdx.x; tile < P.ntiles; tile += gridDim.x, buf ^= 1) {
#ifdef W3P_STAMP
    const bool st_ = tile == (int)blockIdx.x + 3 * (int)gridDim.x && tid < 64;
    W3P_ST(4, st_);
#endif
    if constexpr (GEO > 0) {
      constexpr W3Geo Gm = w3_geo(GEO);
      constexpr uint32_t kstep = 2 * (16 >> Gm.lgTW) * Gm.PW * Gm.CC, bstep = 2 * 16 * 32 * Gm.ntw;
      constexpr uint32_t xloc = 2 * Gm.PP * Gm.CC, dloc = 2 * IG_TILE_M * 32 * Gm.ntw;
      const uint32_t bo = 2 * buf * buf_u16;
      uint32_t xab[MAXS][2], xbb[2];
#pragma unroll
      for (int s = 0; s < MAXS; ++s) { xab[s][0] = xa[s][0] + bo; xab[s][1] = xa[s][1] + bo; }
      xbb[0] = xb[0] + bo; xbb[1] = xb[1] + bo;
#pragma unroll
      for (int kpi = 0; kpi < IG_TILE_M / 16; ++kpi) {
        const uint32_t ko = kpi * kstep, kb = kpi * bstep;
        w3_bf16x8 ah[2][SG], al[2][SG], bh, bl;
#pragma unroll
        for (int g = 0; g <= NG; ++g) {
          if (g < NG) {
#pragma unroll
            for (int j = 0; j < SG; ++j) {
              const int sl = g * SG + j;
              if (sl < MAXS) {
                ah[g & 1][j] = w3_frag(xab[sl][0] + ko, xab[sl][1] + ko);
                al[g & 1][j] = w3_frag(xab[sl][0] + (ko + xloc), xab[sl][1] + (ko + xloc));
                if (sl == 0) {
                  bh = w3_frag(xbb[0] + kb, xbb[1] + kb);
                  bl = w3_frag(xbb[0] + (kb + dloc), xbb[1] + (kb + dloc));
                }
              }
            }
          }
          __builtin_amdgcn_sched_barrier(0);
          if (g > 0) {
#pragma unroll
            for (int j = 0; j < SG; ++j) {
              const int sl = (g - 1) * SG + j;
              if (sl < MAXS) {
                acc[sl] = w3_mfma<ABF>(ah[(g - 1) & 1][j], al[(g - 1) & 1][j], bh, bl, acc[sl]);
              }
            }
            __builtin_amdgcn_sched_barrier(0);
          }
        }
      }
    } else {
      const uint32_t kstep = 2 * (16 >> P.lgTW) * PW * CC;
      const uint32_t bstep = 2 * 16 * DYW;
      uint32_t ko = 2 * buf * buf_u16, kb = ko;
#pragma unroll 1
      for (int kp = 0; kp < IG_TILE_M; kp += 16, ko += kstep, kb += bstep) {
        w3_bf16x8 ah[2][SG], al[2][SG], bh, bl;
#pragma unroll
        for (int g = 0; g <= NG; ++g) {
          if (g < NG) {
#pragma unroll
            for (int j = 0; j < SG; ++j) {
              const int sl = g * SG + j;
              if (sl < MAXS) {
                ah[g & 1][j] = w3_frag(xa[sl][0] + ko, xa[sl][1] + ko);
                al[g & 1][j] = w3_frag(xa[sl][0] + ko + xlo, xa[sl][1] + ko + xlo);
                if (sl == 0) {
                  bh = w3_frag(xb[0] + kb, xb[1] + kb);
                  bl = w3_frag(xb[0] + kb + dlo, xb[1] + kb + dlo);
                }
              }
            }
          }
          __builtin_amdgcn_sched_barrier(0);
          if (g > 0) {
#pragma unroll
            for (int j = 0; j < SG; ++j) {
              const int sl = (g - 1) * SG + j;
              if (sl < MAXS) {
                acc[sl] = w3_mfma<ABF>(ah[(g - 1) & 1][j], al[(g - 1) & 1][j], bh, bl, acc[sl]);
              }
            }
            __builtin_amdgcn_sched_barrier(0);
          }
        }
      }
    }
#ifdef W3P_STAMP
    W3P_ST(5, st_);
#endif
    __syncthreads();
#ifdef W3P_STAMP
    W3P_ST(6, st_);
    if (st_ && lane == 0) {
      unsigned hw_;
      asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hw_));
      w3p_stamp_buf[((blockIdx.z * gridDim.y + blockIdx.y) * gridDim.x + blockIdx.x) % 1024 * 8 + 7] = hw_;
    }
#endif
  }
  const int NPo = gridDim.y * DYW;
#pragma unroll
  for (int s = 0; s < MAXS; ++s) {
    if (valid[s]) {
      const int it = wave + NW * s;
      const int cit = it % P.nct, rr = it / P.nct;
      const int nt = rr % P.ntw, tap = rr / P.ntw;
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        int ci = cz0 + cit * 32 + crow(r, lh), tp = tap;
        if (P.pack2) {
          const int row = crow(r, lh);
          tp = 2 * tap + (row >> 4);
          ci = row & 15;
          if (tp >= p.ntaps) continue;
        }
        p.part[(((size_t)blockIdx.x * p.ntaps + tp) * p.CINP + ci) * NPo + n0 + nt * 32 + li] = acc[s][r];
      }
    }
  }
}

__global__ void reduce_partials_kernel(const float* __restrict__ part, int G, int ntaps, int KP, int NP, int K,
                                       int N, float* __restrict__ dst, long s_tap, long s_k, long s_n,
                                       int accumulate) {
  const long total = (long)ntaps * KP * NP;
  for (long e = (long)blockIdx.x * blockDim.x + threadIdx.x; e < total; e += (long)gridDim.x * blockDim.x) {
    const int n = (int)(e % NP);
    const long r = e / NP;
    const int k = (int)(r % KP), tap = (int)(r / KP);
    if (n >= N || k >= K) continue;
    float s0 = 0.f, s1 = 0.f, s2 = 0.f, s3 = 0.f;
    int g = 0;
    for (; g + 4 <= G; g += 4) {
      s0 += part[(size_t)g * total + e];
      s1 += part[(size_t)(g + 1) * total + e];
      s2 += part[(size_t)(g + 2) * total + e];
      s3 += part[(size_t)(g + 3) * total + e];
    }
    for (; g < G; ++g) s0 += part[(size_t)g * total + e];
    const float s = (s0 + s1) + (s2 + s3);
    float* d = dst + tap * s_tap + k * s_k + n * s_n;
    *d = accumulate ? *d + s : s;
  }
}

// Small outputs (a 16x16 or 32x32 dW reduced over ~1000 slabs) starve the kernel above of parallelism: one thread
// per output element walks all G slabs.  Here a workgroup owns 64 consecutive elements and its S waves split the
// slabs (wave w takes g = w, w+S, ...); the S partial sums are combined through LDS in a fixed order (deterministic).
__global__ __launch_bounds__(1024) void reduce_partials_split_kernel(
    const float* __restrict__ part, int G, int ntaps, int KP, int NP, int K, int N, float* __restrict__ dst,
    long s_tap, long s_k, long s_n, int accumulate) {
  __shared__ float red[16][64];
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6, S = blockDim.x >> 6;
  const long total = (long)ntaps * KP * NP;
  const long e = (long)blockIdx.x * 64 + lane;
  float s0 = 0.f, s1 = 0.f, s2 = 0.f, s3 = 0.f;
  if (e < total) {
    int g = w;
    for (; g + 7 * S < G; g += 8 * S) {   // eight slabs in flight per lane
      const float a0 = part[(size_t)g * total + e], a1 = part[(size_t)(g + S) * total + e];
      const float a2 = part[(size_t)(g + 2 * S) * total + e], a3 = part[(size_t)(g + 3 * S) * total + e];
      const float a4 = part[(size_t)(g + 4 * S) * total + e], a5 = part[(size_t)(g + 5 * S) * total + e];
      const float a6 = part[(size_t)(g + 6 * S) * total + e], a7 = part[(size_t)(g + 7 * S) * total + e];
      s0 += a0; s1 += a1; s2 += a2; s3 += a3;
      s0 += a4; s1 += a5; s2 += a6; s3 += a7;
    }
    for (; g + 3 * S < G; g += 4 * S) {
      s0 += part[(size_t)g * total + e];
      s1 += part[(size_t)(g + S) * total + e];
      s2 += part[(size_t)(g + 2 * S) * total + e];
      s3 += part[(size_t)(g + 3 * S) * total + e];
    }
    for (; g < G; g += S) s0 += part[(size_t)g * total + e];
  }
  red[w][lane] = (s0 + s1) + (s2 + s3);
  __syncthreads();
  if (w == 0 && e < total) {
    float s = red[0][lane];
    for (int i = 1; i < S; ++i) s += red[i][lane];
    const int n = (int)(e % NP);
    const long r = e / NP;
    const int k = (int)(r % KP), tap = (int)(r / KP);
    if (n < N && k < K) {
      float* d = dst + tap * s_tap + k * s_k + n * s_n;
      *d = accumulate ? *d + s : s;
    }
  }
}

// Up to BSED_REDUCE_MAX_JOBS partial-slab reductions in ONE launch.  A training step ends ~30 contractions with such a
// reduction (weight gradients, bias gradients), each a 10-12 us launch for microseconds of work: the results are only
// needed by the optimizer (or the gradient all-reduce), so the host queues them and flushes the queue once.  A
// workgroup (16 waves) owns 64 * (16 / S) consecutive elements of one job and its waves split that job's slabs S ways
// (S = 16 for the smallest outputs); partial sums are combined through LDS in a fixed order: deterministic.
struct RpJob {
  const float* part; float* dst;
  int G, KP, NP, K, N, S, accumulate, wg0;   // wg0 = first workgroup of this job
  long total, s_tap, s_k, s_n;
};
struct RpBatch { RpJob j[BSED_REDUCE_MAX_JOBS]; int njobs; };

__global__ __launch_bounds__(1024) void reduce_partials_batch_kernel(const RpBatch Bt) {
  __shared__ float red[16][64];
  int ji = 0;
  for (int k = 1; k < Bt.njobs; ++k)
    if ((int)blockIdx.x >= Bt.j[k].wg0) ji = k;
  const RpJob& J = Bt.j[ji];
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6, S = J.S;
  const int sub = w / S, ws = w % S;             // 16 / S element groups per workgroup, S slab shares each
  const long total = J.total;
  const long e = ((long)(blockIdx.x - J.wg0) * (16 / S) + sub) * 64 + lane;
  float s0 = 0.f, s1 = 0.f, s2 = 0.f, s3 = 0.f;
  if (e < total) {
    const float* part = J.part;
    const int G = J.G;
    int g = ws;
    for (; g + 7 * S < G; g += 8 * S) {   // eight slabs in flight per lane
      const float a0 = part[(size_t)g * total + e], a1 = part[(size_t)(g + S) * total + e];
      const float a2 = part[(size_t)(g + 2 * S) * total + e], a3 = part[(size_t)(g + 3 * S) * total + e];
      const float a4 = part[(size_t)(g + 4 * S) * total + e], a5 = part[(size_t)(g + 5 * S) * total + e];
      const float a6 = part[(size_t)(g + 6 * S) * total + e], a7 = part[(size_t)(g + 7 * S) * total + e];
      s0 += a0; s1 += a1; s2 += a2; s3 += a3;
      s0 += a4; s1 += a5; s2 += a6; s3 += a7;
    }
    for (; g + 3 * S < G; g += 4 * S) {   // (same assignment of slabs to the four sums as bsed_reduce_partials:
      s0 += part[(size_t)g * total + e];  //  a queued and an immediate reduction give the same bits)
      s1 += part[(size_t)(g + S) * total + e];
      s2 += part[(size_t)(g + 2 * S) * total + e];
      s3 += part[(size_t)(g + 3 * S) * total + e];
    }
    for (; g < G; g += S) s0 += part[(size_t)g * total + e];
  }
  red[w][lane] = (s0 + s1) + (s2 + s3);
  __syncthreads();
  if (ws == 0 && e < total) {
    float s = red[w][lane];
    for (int i = 1; i < S; ++i) s += red[w + i][lane];
    const int n = (int)(e % J.NP);
    const long r = e / J.NP;
    const int k = (int)(r % J.KP), tap = (int)(r / J.KP);
    if (n < J.N && k < J.K) {
      float* d = J.dst + tap * J.s_tap + k * J.s_k + n * J.s_n;
      *d = J.accumulate ? *d + s : s;
    }
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
    static BsedLdsOnce once;                                                                         \
    BSED_HIP(bsed_max_lds(once, (const void*)igemm_kernel<KC, BN, E>));                                                                                                 \
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
  P.pw_magic = ((1 << 20) + P.PW - 1) / P.PW;
  for (int pos = 0; pos < P.PP; ++pos)
    BSED_CHECK_ARG(((pos * P.pw_magic) >> 20) == pos / P.PW, "bsed_igemm: internal: magic division fails for PW=%d", P.PW);
  size_t fl = (size_t)P.b_off + (size_t)KC * BN;
  if (d.epilogue == EPI_GLU_POOL) {
    BSED_CHECK_ARG((d.ph == 1 || d.ph == 2) && (d.pw == 1 || d.pw == 2) && d.TH % d.ph == 0 && d.TW % d.pw == 0,
                   "bsed_igemm: pooling windows must be 1 or 2 and divide the tile");
    BSED_CHECK_ARG(d.Hp == d.H / d.ph && d.Wp == d.W / d.pw, "bsed_igemm: bad pooled shape");
    fl = std::max(fl, (size_t)IG_TILE_M * (BN + 1));
  }
  if (d.epilogue == EPI_GLU_POOL || d.epilogue == EPI_GLU_BWD)
    BSED_CHECK_ARG(d.e_src && d.e_scale && d.e_shift && d.e_pitch >= d.N, "bsed_igemm: GLU epilogue needs e_src/e_scale/e_shift");
  if (d.epilogue == EPI_GLU_BWD)
    BSED_CHECK_ARG(d.e_dpool && d.out2 && d.stats && (d.ph == 1 || d.ph == 2) && (d.pw == 1 || d.pw == 2) &&
                       d.Hp == d.H / d.ph && d.Wp == d.W / d.pw,
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

template <int MAXS, int NW>
static int launch_wgrad(const WgradParams& P, dim3 grid, size_t smem, hipStream_t s) {
  static BsedLdsOnce once;
  BSED_HIP(bsed_max_lds(once, (const void*)wgrad_kernel<MAXS, NW>));
  hipLaunchKernelGGL((wgrad_kernel<MAXS, NW>), grid, dim3(NW * 64), smem, s, P);
  BSED_LAUNCH_CHECK();
  return BSED_OK;
}

// shape checks + derived launch geometry shared by bsed_wgrad and bsed_wgrad_auto_g
static int wgrad_prepare(const BsedWgradDesc* desc, WgradParams& P, size_t& smem, dim3& grid_yz, int mode3 = 0) {
  BSED_CHECK_ARG(desc, "bsed_wgrad: null descriptor");
  P.d = *desc;
  BsedWgradDesc& d = P.d;
  BSED_CHECK_ARG(d.NB > 0 && d.H > 0 && d.W > 0 && d.CIN > 0 && d.N > 0, "bsed_wgrad: bad shape");
  BSED_CHECK_ARG(d.TH * d.TW == IG_TILE_M, "bsed_wgrad: TH*TW must be 128");
  P.lgTW = ilog2_exact(d.TW);
  BSED_CHECK_ARG(P.lgTW >= 0 && d.W % d.TW == 0, "bsed_wgrad: TW must be a power of two dividing W");
  BSED_CHECK_ARG(d.ntaps >= 1 && d.ntaps <= 9, "bsed_wgrad: ntaps must be in 1..9");
  for (int t = 0; t < d.ntaps; ++t)
    BSED_CHECK_ARG(abs(d.dh[t]) <= d.hh && abs(d.dw[t]) <= d.hw, "bsed_wgrad: tap %d outside the halo", t);
  BSED_CHECK_ARG(d.CIN % 4 == 0 && d.N % 4 == 0, "bsed_wgrad: CIN and N must be multiples of 4");
  BSED_CHECK_ARG(d.CINP % 32 == 0 && d.CINP >= d.CIN && d.NP % 32 == 0 && d.NP >= d.N, "bsed_wgrad: bad padding");
  BSED_CHECK_ARG(d.in_pitch >= d.CIN && d.in_pitch % 4 == 0 && d.dy_pitch >= d.N && d.dy_pitch % 4 == 0, "bsed_wgrad: bad pitch");
  BSED_CHECK_ARG((d.bn_y == nullptr) == (d.bn_coef == nullptr) && (d.bn_y == nullptr) == (d.bn_mean == nullptr),
                 "bsed_wgrad: bn_y, bn_coef and bn_mean come together (BatchNorm backward applied on load) or not at all");
  BSED_CHECK_ARG(d.bn_y || !d.dy_out, "bsed_wgrad: dy_out only goes with bn_y");
  BSED_CHECK_ARG(!d.bn_y || mode3, "bsed_wgrad: BatchNorm backward on load is built into the split-fp32 kernels (bsed_wgrad3) only");
  BSED_CHECK_ARG(!d.dy_out || (d.dy_out != d.dy && d.dy_out != d.bn_y), "bsed_wgrad: dy_out must not alias dy or bn_y "
                 "(other workgroups still read them)");
  BSED_CHECK_ARG((d.bn_y == nullptr) == (d.bn_coef == nullptr) && (d.bn_y == nullptr) == (d.bn_mean == nullptr),
                 "bsed_wgrad: bn_y, bn_coef and bn_mean come together (BatchNorm backward applied on load) or not at all");
  BSED_CHECK_ARG(d.bn_y || !d.dy_out, "bsed_wgrad: dy_out only goes with bn_y");
  BSED_CHECK_ARG(!d.bn_y || mode3, "bsed_wgrad: BatchNorm backward on load is built into the split-fp32 kernels (bsed_wgrad3) only");
  BSED_CHECK_ARG(!d.dy_out || (d.dy_out != d.dy && d.dy_out != d.bn_y), "bsed_wgrad: dy_out must not alias dy or bn_y "
                 "(other workgroups still read them)");
  d.tilesH = ceil_div(d.H, d.TH);
  d.tilesW = d.W / d.TW;
  P.PW = d.TW + 2 * d.hw;
  P.PH = d.TH + 2 * d.hh;
  P.PP = P.PW * P.PH;
  // input channels are contracted in chunks of CC per workgroup (grid.z): 64 for the 9-tap convolutions so that
  // two workgroups fit in a CU's LDS and one's tile load overlaps the other's MFMAs, 128 for the 1-tap forms
  // (A/B knob: BSED_WGRAD3_1TAP_PIPE=1 gives the split-fp32 1-tap forms the multi-tap geometry -- 64-channel chunks, an
  //  80 KB tile -- so that the producer / consumer kernel with its two tile buffers can take them)
  static const bool onetap_pipe = getenv("BSED_WGRAD3_1TAP_PIPE") && getenv("BSED_WGRAD3_1TAP_PIPE")[0] == '1';
  const bool wide1 = d.ntaps == 1 && !(mode3 && onetap_pipe);
  P.CC = 32;
  for (int cand = (wide1 ? 128 : 64); cand >= 32; cand >>= 1)
    if (d.CINP % cand == 0) { P.CC = cand; break; }
  // tall narrow tiles (W = 2: 66 x 4 halo patch) would leave a single workgroup per CU with 64-channel chunks
  if (mode3 && d.ntaps > 1 && P.CC > 32 && (size_t)P.PP * P.CC * 4 + (size_t)W3_DY_BYTES > 80 * 1024) P.CC = 32;
  BSED_CHECK_ARG(d.CINP % P.CC == 0 && d.CINP / P.CC <= 65535, "bsed_wgrad: CINP must be a multiple of 32");
  P.lgc4 = ilog2_exact(P.CC / 4);
  P.nct = P.CC / 32;
  P.pack2 = (d.CIN <= 16 && d.CINP == 32 && d.ntaps > 1) ? 1 : 0;
  P.dy_off = mode3 ? P.PP * P.CC : ((P.PP * (P.CC + 1) + 3) & ~3);  // in 4-byte words
  P.pw_magic = ((1 << 20) + P.PW - 1) / P.PW;
  for (int pos = 0; pos < P.PP; ++pos)
    BSED_CHECK_ARG(((pos * P.pw_magic) >> 20) == pos / P.PW, "bsed_wgrad: internal: magic division fails for PW=%d", P.PW);
  // 1-tap contractions have one MFMA per LDS pair: widen the dy tile to 128 channels so the activation tile is
  // staged once per 4 output tiles (4x fewer HBM/L2 re-reads of `in`) when it fits in LDS
  // (the 9-tap forms get 2 dy tiles when two workgroups still fit in one CU's LDS: the activation patch is then
  // staged once per 64 output channels and the per-CU load rate stops being the limiter)
  P.ntw = 1;
  for (int cand = 4; cand >= 2; cand >>= 1) {
    const size_t need = mode3 ? (size_t)P.dy_off * 4 + (size_t)W3_DY_BYTES * cand
                              : ((size_t)P.dy_off + IG_TILE_M * 32 * cand) * sizeof(float);
    const size_t budget = wide1 ? 160 * 1024 : 80 * 1024;
    const int tap_items = P.pack2 ? (d.ntaps + 1) / 2 : d.ntaps;
    if (d.NP % (32 * cand) == 0 && tap_items * P.nct * cand <= 36 && need <= budget) { P.ntw = cand; break; }
  }
  smem = mode3 ? (size_t)P.dy_off * 4 + (size_t)W3_DY_BYTES * P.ntw
               : ((size_t)P.dy_off + IG_TILE_M * 32 * P.ntw) * sizeof(float);
  BSED_CHECK_ARG(smem <= 160 * 1024, "bsed_wgrad: tile needs %zu B of LDS", smem);
  const long ntiles = (long)d.NB * d.tilesH * d.tilesW;
  BSED_CHECK_ARG(ntiles < (1L << 31), "bsed_wgrad: too many tiles");
  P.ntiles = (int)ntiles;
  grid_yz = dim3(1, d.NP / (32 * P.ntw), d.CINP / P.CC);
  return BSED_OK;
}

// number of partial slabs G such that G x (n tiles) x (channel chunks) is about three workgroups per CU
extern "C" int bsed_wgrad_auto_g(const BsedWgradDesc* desc) {
  WgradParams P;
  size_t smem;
  dim3 gyz;
  if (wgrad_prepare(desc, P, smem, gyz) != BSED_OK) return -1;
  // exactly one resident round of workgroups (2 per CU when two tiles fit in LDS): no partial tail round
  const long slots = smem <= 80 * 1024 ? 512 : 256;
  const long want = std::max<long>(1, slots / ((long)gyz.y * gyz.z));
  return (int)std::max<long>(1, std::min<long>(want, P.ntiles));
}

// which template instance bsed_wgrad will launch for this shape: MAXS * 16 + NW (profiling / bench labels)
static int wgrad_variant(const WgradParams& P) {
  const BsedWgradDesc& d = P.d;
  const int nitems = (P.pack2 ? (d.ntaps + 1) / 2 : d.ntaps) * P.nct * P.ntw;
  if (nitems >= 16 && d.ntaps == 1) {
    const int slots8 = ceil_div(nitems, 8);
    if (slots8 <= 2) return 2 * 16 + 8;
    if (slots8 <= 3) return 3 * 16 + 8;
    if (slots8 <= 5) return 5 * 16 + 8;
  }
  const int slots = ceil_div(nitems, 4);
  const int m = slots <= 1 ? 1 : slots <= 2 ? 2 : slots <= 3 ? 3 : slots <= 5 ? 5 : 9;
  return m * 16 + 4;
}

extern "C" int bsed_wgrad_variant(const BsedWgradDesc* desc) {
  WgradParams P;
  size_t smem;
  dim3 gyz;
  if (wgrad_prepare(desc, P, smem, gyz) != BSED_OK) return -1;
  return wgrad_variant(P);
}

extern "C" int bsed_wgrad(const BsedWgradDesc* desc, void* stream) {
  WgradParams P;
  size_t smem;
  dim3 gyz;
  int rc = wgrad_prepare(desc, P, smem, gyz);
  if (rc != BSED_OK) return rc;
  BsedWgradDesc& d = P.d;
  BSED_CHECK_ARG(d.in && d.dy && d.part, "bsed_wgrad: null tensor");
  BSED_CHECK_ARG(d.G > 0 && d.G <= P.ntiles, "bsed_wgrad: G (%d) must be in 1..%d tiles", d.G, P.ntiles);
  const int nitems = (P.pack2 ? (d.ntaps + 1) / 2 : d.ntaps) * P.nct * P.ntw;
  dim3 grid((unsigned)d.G, gyz.y, gyz.z);
  hipStream_t s = (hipStream_t)stream;
  if (nitems >= 16 && d.ntaps == 1) {  // 1-tap forms: 8 waves per staged tile (measured +55 %); 9-tap forms are faster with 4
    const int slots8 = ceil_div(nitems, 8);
    if (slots8 <= 2) return launch_wgrad<2, 8>(P, grid, smem, s);
    if (slots8 <= 3) return launch_wgrad<3, 8>(P, grid, smem, s);
    if (slots8 <= 5) return launch_wgrad<5, 8>(P, grid, smem, s);
  }
  const int slots = ceil_div(nitems, 4);
  if (slots <= 1) return launch_wgrad<1, 4>(P, grid, smem, s);
  if (slots <= 2) return launch_wgrad<2, 4>(P, grid, smem, s);
  if (slots <= 3) return launch_wgrad<3, 4>(P, grid, smem, s);
  if (slots <= 5) return launch_wgrad<5, 4>(P, grid, smem, s);
  if (slots <= 9) return launch_wgrad<9, 4>(P, grid, smem, s);
  bsed_set_error("bsed_wgrad: %d work items per workgroup exceed the 36 supported", nitems);
  return BSED_ERR_ARG;
}


template <int MAXS, int NW, bool BS>
static int launch_wgrad3_bs(const WgradParams& P, dim3 grid, size_t smem, hipStream_t s) {
  static BsedLdsOnce once, onceb;
  if (P.d.act_bf16) {
    BSED_HIP(bsed_max_lds(onceb, (const void*)wgrad3_kernel<MAXS, NW, BS, 1>));
    hipLaunchKernelGGL((wgrad3_kernel<MAXS, NW, BS, 1>), grid, dim3(NW * 64), smem, s, P);
  } else {
    BSED_HIP(bsed_max_lds(once, (const void*)wgrad3_kernel<MAXS, NW, BS, 0>));
    hipLaunchKernelGGL((wgrad3_kernel<MAXS, NW, BS, 0>), grid, dim3(NW * 64), smem, s, P);
  }
  BSED_LAUNCH_CHECK();
  return BSED_OK;
}

template <int MAXS, int NW>
static int launch_wgrad3(const WgradParams& P, dim3 grid, size_t smem, hipStream_t s) {
  if (NW % (P.nct * P.ntw) == 0) return launch_wgrad3_bs<MAXS, NW, true>(P, grid, smem, s);
  return launch_wgrad3_bs<MAXS, NW, false>(P, grid, smem, s);
}

template <int MAXS, int GEO>
static int launch_wgrad3p_geo(const WgradParams& P, dim3 grid, size_t smem, hipStream_t s) {
  static BsedLdsOnce once, onceb;
  if (P.d.act_bf16) {
    BSED_HIP(bsed_max_lds(onceb, (const void*)wgrad3p_kernel<MAXS, GEO, 1>));
    hipLaunchKernelGGL((wgrad3p_kernel<MAXS, GEO, 1>), grid, dim3(512), 2 * smem, s, P);
  } else {
    BSED_HIP(bsed_max_lds(once, (const void*)wgrad3p_kernel<MAXS, GEO, 0>));
    hipLaunchKernelGGL((wgrad3p_kernel<MAXS, GEO, 0>), grid, dim3(512), 2 * smem, s, P);
  }
  BSED_LAUNCH_CHECK();
  return BSED_OK;
}

// which compile-time geometry (w3_geo) matches this launch, 0 = none
static int wgrad3p_geo(const WgradParams& P) {
  for (int g = 1; g <= W3_NGEO; ++g) {
    const W3Geo G = w3_geo(g);
    if (P.CC == G.CC && P.lgTW == G.lgTW && P.PW == G.PW && P.PP == G.PP && P.ntw == G.ntw) return g;
  }
  return 0;
}

template <int MAXS>
static int launch_wgrad3p(const WgradParams& P, dim3 grid, size_t smem, hipStream_t s) {
  const int g = getenv("BSED_WGRAD3_NOGEO") ? 0 : wgrad3p_geo(P);
  if (MAXS == 9 && g == 1) return launch_wgrad3p_geo<9, 1>(P, grid, smem, s);
  if (MAXS == 9 && g == 2) return launch_wgrad3p_geo<9, 2>(P, grid, smem, s);
  if (MAXS == 5 && g == 3) return launch_wgrad3p_geo<5, 3>(P, grid, smem, s);
  if (MAXS == 5 && g == 4) return launch_wgrad3p_geo<5, 4>(P, grid, smem, s);
  if (MAXS == 5 && g == 5) return launch_wgrad3p_geo<5, 5>(P, grid, smem, s);
  if (MAXS == 3 && g == 6) return launch_wgrad3p_geo<3, 6>(P, grid, smem, s);
  return launch_wgrad3p_geo<MAXS, 0>(P, grid, smem, s);
}

// the pipelined kernel takes the multi-tap shapes whose two tile buffers fit in LDS and whose per-thread prefetch fits
// its register arrays; everything else (1-tap forms, tall narrow patches) stays on wgrad3_kernel
static bool wgrad3_pipelined(const WgradParams& P, size_t smem) {
  const int nitems = (P.pack2 ? (P.d.ntaps + 1) / 2 : P.d.ntaps) * P.nct * P.ntw;
  // fewest (tap, chunk) items the pipelined kernel takes: 4, i.e. the 16 -> 32 channel layer (5 items: two taps per MFMA
  // tile) included.  With run-time geometry the pipelined kernel lost to wgrad3_kernel there with fp32 activations (0.78
  // vs 0.69 ms; bf16: 0.41 vs 0.65); with the layer's geometry compiled in (w3_geo(6)) it wins in both modes: 0.60 ms
  // fp32, 0.28 ms bf16.  BSED_WGRAD3_PIPE_MIN: A/B knob (8 = the old choice)
  static const int pipe_env = getenv("BSED_WGRAD3_PIPE_MIN") ? atoi(getenv("BSED_WGRAD3_PIPE_MIN")) : 0;
  const int pipe_min = pipe_env > 0 ? pipe_env : 4;
  static const bool onetap_pipe = getenv("BSED_WGRAD3_1TAP_PIPE") && getenv("BSED_WGRAD3_1TAP_PIPE")[0] == '1';
  if (P.d.ntaps == 1 && !onetap_pipe) return false;
  return nitems > (P.d.ntaps == 1 ? 3 : pipe_min) &&
         P.d.H < 16384 && P.d.W < 16384 && 2 * smem <= 160 * 1024 && 4 % (P.nct * P.ntw) == 0 && nitems <= 36 &&
         P.PP * (P.CC / 4) <= W3P_UX * 256 && IG_TILE_M * 8 * P.ntw <= W3P_UD * 256 && !getenv("BSED_WGRAD3_NOPIPE");
}

// the streaming 1-tap kernel takes the shapes whose slabs tile into 128 x 128 blocks (BSED_WGRAD1=0: A/B against
// wgrad3_kernel)
static bool wgrad1_streaming(const WgradParams& P) {
  static const bool off = getenv("BSED_WGRAD1") && getenv("BSED_WGRAD1")[0] == '0';
  const BsedWgradDesc& d = P.d;
  return !off && d.ntaps == 1 && P.CC == 128 && P.ntw == 4 && d.CINP % 128 == 0 && d.NP % 128 == 0 && !d.bn_y &&
         (long)d.NB * d.H * d.W < (1L << 31) - 4 * 65536;
}

extern "C" int bsed_wgrad3_auto_g(const BsedWgradDesc* desc) {
  WgradParams P;
  size_t smem;
  dim3 gyz;
  if (wgrad_prepare(desc, P, smem, gyz, 1) != BSED_OK) return -1;
  long slots = wgrad3_pipelined(P, smem) ? 256 : smem <= 80 * 1024 ? 512 : 256;
  // small tiles (conv1 16 -> 32 channel gradient: 39 KB, 136 registers): a third workgroup per CU fits and hides more of
  // the stage / barrier / MFMA serialisation: 0.956 -> 0.753 ms; a fourth (the kernel pinned to 128 registers) 0.651 ms
  // (BSED_WGRAD3_SLOTS_SMALL for A/B runs)
  static const int small_slots = getenv("BSED_WGRAD3_SLOTS_SMALL") ? atoi(getenv("BSED_WGRAD3_SLOTS_SMALL")) : 1024;
  if (!wgrad3_pipelined(P, smem) && smem <= 52 * 1024 && wgrad_variant(P) / 16 <= 2) slots = small_slots;
  long want = std::max<long>(1, slots / ((long)gyz.y * gyz.z));
  // XCD affinity: workgroup (x, y, z) has linear id x + G * (y + gy * z) and lands on XCD id % 8.  The gy * gz
  // workgroups of one tile sequence x read the same activation / dy tiles at about the same time; with G a multiple of
  // 8 they share an XCD, hence its L2, and the re-reads stop going to HBM (BSED_WGRAD3_G8=0 disables, for A/B runs)
  const char* g8 = getenv("BSED_WGRAD3_G8");
  if (gyz.y * gyz.z > 1 && (want % 8) * 16 <= want && !(g8 && g8[0] == '0')) want -= want % 8;  // <= 6 % fewer workgroups
  return (int)std::max<long>(1, std::min<long>(want, P.ntiles));
}

extern "C" int bsed_wgrad3_variant(const BsedWgradDesc* desc) {
  WgradParams P;
  size_t smem;
  dim3 gyz;
  if (wgrad_prepare(desc, P, smem, gyz, 1) != BSED_OK) return -1;
  if (wgrad1_streaming(P)) return (1 << 13) | ((P.d.dh[0] != 0 || P.d.dw[0] != 0) ? 1 : 0);  // wgrad1_kernel<SHIFT>
  const int v = wgrad_variant(P);
  if (wgrad3_pipelined(P, smem)) {  // NW field 1 = wgrad3p_kernel<MAXS, GEO>, GEO in bits 8..11
    const int maxs = v / 16 <= 3 ? 3 : (v / 16 <= 5 ? 5 : 9), g = getenv("BSED_WGRAD3_NOGEO") ? 0 : wgrad3p_geo(P);
    const bool built = (maxs == 9 && (g == 1 || g == 2)) || (maxs == 5 && g >= 3 && g <= 5) || (maxs == 3 && g == 6);
    return maxs * 16 + 1 + ((built ? g : 0) << 8);
  }
  return v | (((v % 16) % (P.nct * P.ntw) == 0) ? 1 << 12 : 0);  // bit 12 = the BS template argument
}

extern "C" int bsed_wgrad3(const BsedWgradDesc* desc, void* stream) {
  WgradParams P;
  size_t smem;
  dim3 gyz;
  int rc = wgrad_prepare(desc, P, smem, gyz, 1);
  if (rc != BSED_OK) return rc;
  BsedWgradDesc& d = P.d;
  BSED_CHECK_ARG(d.in && d.dy && d.part, "bsed_wgrad3: null tensor");
  BSED_CHECK_ARG(d.G > 0 && d.G <= P.ntiles, "bsed_wgrad3: G (%d) must be in 1..%d tiles", d.G, P.ntiles);
  dim3 grid((unsigned)d.G, gyz.y, gyz.z);
  hipStream_t s = (hipStream_t)stream;
  if (wgrad1_streaming(P)) {
    const bool shifted = d.dh[0] != 0 || d.dw[0] != 0;
    static BsedLdsOnce once0, once1, once0b, once1b;
    if (d.act_bf16 && shifted) {
      BSED_HIP(bsed_max_lds(once1b, (const void*)wgrad1_kernel<true, 1>));
      hipLaunchKernelGGL((wgrad1_kernel<true, 1>), grid, dim3(W1_THREADS), 2 * W1_STAGE, s, P);
    } else if (d.act_bf16) {
      BSED_HIP(bsed_max_lds(once0b, (const void*)wgrad1_kernel<false, 1>));
      hipLaunchKernelGGL((wgrad1_kernel<false, 1>), grid, dim3(W1_THREADS), 2 * W1_STAGE, s, P);
    } else if (shifted) {
      BSED_HIP(bsed_max_lds(once1, (const void*)wgrad1_kernel<true, 0>));
      hipLaunchKernelGGL((wgrad1_kernel<true, 0>), grid, dim3(W1_THREADS), 2 * W1_STAGE, s, P);
    } else {
      BSED_HIP(bsed_max_lds(once0, (const void*)wgrad1_kernel<false, 0>));
      hipLaunchKernelGGL((wgrad1_kernel<false, 0>), grid, dim3(W1_THREADS), 2 * W1_STAGE, s, P);
    }
    BSED_LAUNCH_CHECK();
    return BSED_OK;
  }
  const int v = wgrad_variant(P), maxs = v / 16, nw = v % 16;
  if (wgrad3_pipelined(P, smem)) {
    if (maxs <= 3) return launch_wgrad3p<3>(P, grid, smem, s);
    if (maxs <= 5) return launch_wgrad3p<5>(P, grid, smem, s);
    return launch_wgrad3p<9>(P, grid, smem, s);
  }
  if (nw == 8) {
    if (maxs == 2) return launch_wgrad3<2, 8>(P, grid, smem, s);
    if (maxs == 3) return launch_wgrad3<3, 8>(P, grid, smem, s);
    return launch_wgrad3<5, 8>(P, grid, smem, s);
  }
  if (maxs == 1) return launch_wgrad3<1, 4>(P, grid, smem, s);
  if (maxs == 2) return launch_wgrad3<2, 4>(P, grid, smem, s);
  if (maxs == 3) return launch_wgrad3<3, 4>(P, grid, smem, s);
  if (maxs == 5) return launch_wgrad3<5, 4>(P, grid, smem, s);
  return launch_wgrad3<9, 4>(P, grid, smem, s);
}

extern "C" int bsed_reduce_partials(const float* part, int G, int ntaps, int KP, int NP, int K, int N, float* dst,
                                    long s_tap, long s_k, long s_n, int accumulate, void* stream) {
  BSED_CHECK_ARG(part && dst && G > 0 && ntaps > 0 && KP >= K && NP >= N && K > 0 && N > 0, "bsed_reduce_partials: bad argument");
  const long total = (long)ntaps * KP * NP;
  const long wgs = ceil_div(total, 64);
  if (wgs < 1024 && G >= 8) {  // too few elements to fill the chip: split the slabs over the waves of a workgroup
    int S = 1;
    while (S < 16 && wgs * S < 1024 && 4 * S <= G) S *= 2;
    hipLaunchKernelGGL(reduce_partials_split_kernel, dim3((unsigned)wgs), dim3(64 * S), 0, (hipStream_t)stream, part,
                       G, ntaps, KP, NP, K, N, dst, s_tap, s_k, s_n, accumulate);
    BSED_LAUNCH_CHECK();
    return BSED_OK;
  }
  hipLaunchKernelGGL(reduce_partials_kernel, dim3((unsigned)std::min<long>(ceil_div(total, 256), 4096)), dim3(256), 0,
                     (hipStream_t)stream, part, G, ntaps, KP, NP, K, N, dst, s_tap, s_k, s_n, accumulate);
  BSED_LAUNCH_CHECK();
  return BSED_OK;
}

extern "C" int bsed_reduce_partials_batch(const BsedReduceJob* jobs, int njobs, void* stream) {
  BSED_CHECK_ARG(jobs && njobs > 0 && njobs <= BSED_REDUCE_MAX_JOBS, "bsed_reduce_partials_batch: 1..%d jobs",
                 BSED_REDUCE_MAX_JOBS);
  RpBatch Bt;
  Bt.njobs = njobs;
  long wg = 0;
  for (int i = 0; i < njobs; ++i) {
    const BsedReduceJob& q = jobs[i];
    BSED_CHECK_ARG(q.part && q.dst && q.G > 0 && q.ntaps > 0 && q.KP >= q.K && q.NP >= q.N && q.K > 0 && q.N > 0,
                   "bsed_reduce_partials_batch: bad job %d", i);
    for (int k = 0; k < i; ++k)
      BSED_CHECK_ARG(jobs[k].dst != q.dst, "bsed_reduce_partials_batch: jobs %d and %d write the same destination "
                     "(flush between them)", k, i);
    RpJob& J = Bt.j[i];
    J.part = q.part; J.dst = q.dst; J.G = q.G; J.KP = q.KP; J.NP = q.NP; J.K = q.K; J.N = q.N;
    J.accumulate = q.accumulate; J.s_tap = q.s_tap; J.s_k = q.s_k; J.s_n = q.s_n;
    J.total = (long)q.ntaps * q.KP * q.NP;
    const long chunks = ceil_div(J.total, 64);   // 64-element groups
    int S = 1;
    if (chunks < 1024 && q.G >= 8)   // the split rule of bsed_reduce_partials
      while (S < 16 && chunks * S < 1024 && 4 * S <= q.G) S *= 2;
    J.S = S;
    J.wg0 = (int)wg;
    wg += ceil_div(chunks, 16 / S);
    BSED_CHECK_ARG(wg < (1L << 31), "bsed_reduce_partials_batch: too many workgroups");
  }
  hipLaunchKernelGGL(reduce_partials_batch_kernel, dim3((unsigned)wg), dim3(1024), 0, (hipStream_t)stream, Bt);
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
