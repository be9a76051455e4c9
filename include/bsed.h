/* bsed.h -- C ABI of libbsed.so, the MI355X (gfx950) drop-in for the mel + CRNN train-step hot path
 * of fumchin/bird-sound-event-detecion.
 *
 * The reference is pure Python/PyTorch and has no FFI of its own: the drop-in
 * boundary is its Python API (CRNN / Predictor / preprocess / train_mt / update_ema_variables /
 * get_predictions).  The package `bird-sound-event-detecion_amd` mirrors that API and binds THIS
 * library with ctypes; each entry point below names the reference code it replaces.
 *
 * Conventions
 *   - plain pointers and sizes only; every pointer is DEVICE memory owned by the caller unless the
 *     comment says "host";  tensors are contiguous fp32 unless a pitch is given.
 *   - activations are NHWC: (N, H=time, W=freq, C).  A reference (B,1,T,F) input is the same bytes.
 *   - every call is asynchronous on `stream` (a hipStream_t; NULL = default stream); nothing
 *     allocates, frees or synchronises except the *_create / *_destroy pairs.
 *   - return 0 on success, <0 on failure; bsed_last_error() returns a thread-local message.
 */
#ifndef BSED_H
#define BSED_H
#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

const char* bsed_last_error(void);
/* "gfx950" build tag + ABI version, for the loader's sanity check */
const char* bsed_build_info(void);
int bsed_abi_version(void);

/* ------------------------------------------------------------------------------------------------
 * Mel front end.   replaces preprocess()  (reference src/data/preprocess.py:18-45,
 * src/synth_data/synth_data_preprocess.py:15-42) and the DataLoader transforms
 * AugmentGaussianNoise / ApplyLog / PadOrTrunc (src/data/Transforms.py:74-86,89-139,155-196).
 * ---------------------------------------------------------------------------------------------- */
typedef struct BsedMelCfg {
  int sr;      /* 32000 (reference) or 22050 (BASELINE measurement config) */
  int n_fft;   /* 2048 */
  int hop;     /* 255 */
  int n_mels;  /* 128 */
  float fmin;  /* 0 */
  float fmax;  /* 16000 (must be <= sr/2) */
} BsedMelCfg;

int bsed_mel_plan_create(const BsedMelCfg* cfg /*host*/, void** plan /*host out*/);
int bsed_mel_plan_destroy(void* plan);
int bsed_mel_plan_nnz(const void* plan);                    /* non-zeros of the filterbank */
int bsed_mel_plan_frames_per_wave(const void* plan);        /* 2: bsed_mel_linear runs stft_mel2_kernel (two frames per wave), 1: stft_mel_kernel */
int bsed_mel_num_frames(const void* plan, int n_samples);   /* 1 + n_samples / hop */
/* wav (B, n_samples) -> LINEAR mel amplitude (B, T, n_mels) == preprocess(audio).  Also emits the
 * per-clip max (B) and the per-(clip, band) sum over time of x^2 (B, n_mels), which the dB clamp and
 * the SNR noise need, so neither costs another pass over HBM.  scratch: bsed_mel_scratch_floats(plan, B, n_samples)
 * floats (per-workgroup partial sums of squares, added in fixed order: the result is bitwise repeatable). */
long bsed_mel_scratch_floats(const void* plan, int B, int n_samples);
int bsed_mel_linear(const void* plan, const float* wav, int B, int n_samples, float* mel_lin,
                    float* clip_max, float* bin_sumsq, float* scratch, void* stream);
/* the same two statistics for features that arrive as linear mel (the reference's wav/<name>.npy files) */
int bsed_mel_stats(const float* mel_lin, int B, int T, int n_mels, float* clip_max, float* bin_sumsq, void* stream);
/* noisy = mel + N(0,1) * sqrt(mean_t(mel^2) * 10^(-snr/10))  (AugmentGaussianNoise.gaussian_noise).
 * unit_noise (B,T,n_mels) may inject the N(0,1) draws (parity tests); NULL = Philox(seed). */
int bsed_mel_noise(const float* mel_lin, const float* bin_sumsq, const float* unit_noise, int B, int T,
                   int n_mels, float snr_db, uint64_t seed, float* noisy, float* clip_max_noisy,
                   void* stream);
/* librosa.amplitude_to_db(x, ref=1, amin=1e-5, top_db) per clip + zero pad / truncate to T_out
 * -> (B, T_out, n_mels), i.e. the (B,1,T_out,n_mels) CRNN input. */
int bsed_mel_db(const float* mel_lin, const float* clip_max, int B, int T, int T_out, int n_mels,
                float top_db, float* out_db, void* stream);

/* ------------------------------------------------------------------------------------------------
 * Implicit-GEMM contraction on the fp32 matrix cores (v_mfma_f32_32x32x2_f32).
 *   out[p][n] = sum_{tap,k} in[p + (dh,dw)(tap)][k] * w[tap][k][n]      zero padding outside (H,W)
 * One body, selected by `epilogue`, replaces the stock ATen/cuDNN ops the reference reaches through
 *   nn.Conv2d 3x3 + train-mode BatchNorm2d statistics   (src/models/CNN.py:46-49)       BSED_EPI_STATS
 *   GLU = Linear(C,C)(bn(x)) * sigmoid(bn(x)), Dropout, AvgPool2d (CNN.py:5-16,59-67)  BSED_EPI_GLU_POOL
 *   their backward passes                                                BSED_EPI_GLU_BWD / _ADD_STATS2
 *   nn.GRU input projections x @ W_ih^T and data gradients (src/models/RNN.py:12)      BSED_EPI_PLAIN
 * ---------------------------------------------------------------------------------------------- */
enum {
  BSED_EPI_PLAIN = 0,      /* out = acc + bias                                                        */
  BSED_EPI_STATS = 1,      /* + per-tile (sum, sum of squares) per channel -> stats[tile][2][N]        */
  BSED_EPI_GLU_POOL = 2,   /* out = avgpool(dropout((acc+bias) * sigmoid(e_src*e_scale+e_shift)))      */
  BSED_EPI_GLU_BWD = 3,    /* out = d_lin, out2 = d_res*lin*sig*(1-sig); stats[tile][0][N] = sum d_lin */
  BSED_EPI_ADD_STATS2 = 4  /* out = acc + out2; stats = (sum g, sum g*e_src)                           */
};

typedef struct BsedIgemmDesc {
  const float* in;       /* (NB,H,W,in_pitch) NHWC; CIN <= in_pitch                                    */
  const float* w;        /* packed (ntaps, CIN, NP) by bsed_pack_weight                                */
  const float* bias;     /* (N) or NULL                                                                */
  float* out;            /* (NB,H,W,out_pitch)  [GLU_POOL: (NB,Hp,Wp,out_pitch)]                       */
  float* out2;           /* GLU_BWD: second output; ADD_STATS2: residual input (may alias out)         */
  float* stats;          /* (num_tiles, 2, N) per-tile partial sums                                    */
  const float* a_scale;  /* optional per-CIN affine applied while loading `in` (BatchNorm apply)       */
  const float* a_shift;
  const float* e_src;    /* epilogue side input (NB,H,W,e_pitch): pre-BN conv output                   */
  const float* e_scale;  /* (N) BatchNorm scale/shift of the gate branch                               */
  const float* e_shift;
  const float* e_dpool;  /* GLU_BWD: gradient w.r.t. the pooled output (NB,Hp,Wp,N)                    */
  int in_pitch, out_pitch, e_pitch;
  int NB, H, W, CIN, N, NP;   /* NP = N rounded up to a multiple of 32 (weight row stride)            */
  int TH, TW;                 /* spatial tile, TH*TW == 128, TW a power of two dividing W             */
  int tilesH, tilesW;         /* filled by the library                                                 */
  int hh, hw;                 /* halo rows / cols = max |dh| / |dw|                                    */
  int ntaps, dh[9], dw[9];
  int ph, pw, Hp, Wp;         /* pooling window and pooled extent (floor)                              */
  int epilogue;
  float drop_p;               /* dropout probability of the GLU epilogues (0 = off)                    */
  uint32_t rng_stream;        /* Philox stream id (layer id)                                           */
  uint64_t seed;
  int valid_h, valid_w;       /* 0 = H, W.  Otherwise only output positions (h < valid_h, w < valid_w) are stored and
                               * enter the STATS sums ("valid" convolutions computed on a padded grid: the stride-2
                               * discriminator layers run as 2x2 stride-1 convolutions over space-to-depth input,
                               * whose last row / column of the grid is not an output).  PLAIN / STATS epilogues. */
  int act_bf16;               /* bsed_igemm3n / bsed_igemm3s: 1 = `in` and `out` are bf16 tensors ("bf16" throughput mode:
                               * one bf16 MFMA per product, fp32 accumulation, bias and statistics); 0 = fp32 */
} BsedIgemmDesc;

int bsed_igemm(const BsedIgemmDesc* desc /*host*/, void* stream);
int bsed_igemm_num_tiles(const BsedIgemmDesc* desc /*host*/);

/* Same contraction with split-fp32 operands on the bf16 matrix cores ("bf16x3": a*b ~ a_hi*b_hi + a_hi*b_lo +
 * a_lo*b_hi, fp32 accumulate, ~1e-5 relative; csrc/igemm3.hip).  BSED_EPI_PLAIN / BSED_EPI_STATS only, CIN a
 * multiple of 32; desc->w must come from bsed_pack_weight3 (bf16 hi/lo planes, (ntaps, CIN/32, NP, 64) uint16). */
int bsed_igemm3(const BsedIgemmDesc* desc /*host*/, void* stream);
int bsed_pack_weight3(const float* src, void* dst, int ntaps, int K, int N, int NP, long s_tap, long s_k, long s_n,
                      void* stream);
/* CIN = 16 / 32 variant (the 16 -> 32 channel convolution, data gradients of 32-channel layers): weights of all taps
 * stay in LDS, CIN/16 MFMA K steps per tap,
 * persistent grid of G workgroups per 32 output channels (bsed_igemm3s_auto_g()); desc as for bsed_igemm3 with
 * w = bsed_pack_weight3s table ((NP/32) * ntaps * (K/16) * 2 * 64 * 16 bytes); STATS writes G partial rows (one per workgroup). */
int bsed_pack_weight3s(const float* src, void* dst, int ntaps, int K, int N, int NP, long s_tap, long s_k, long s_n,
                       void* stream);
/* several bsed_pack_weight3 (kind 0) / bsed_pack_weight3s (kind 1) re-layouts in one launch: the weights of a train step
 * are constant until its optimizer update, so every packed copy the step needs can be made at its start (same bits as the
 * single-job entries; the reference has no counterpart: its convolutions read the PyTorch weight tensors directly,
 * src/models/CNN.py:46-47) */
#define BSED_PACK_MAX_JOBS 32
typedef struct BsedPackJob {
  const float* src; void* dst;
  int kind, ntaps, K, N, NP;
  long s_tap, s_k, s_n;
} BsedPackJob;
int bsed_pack_weights_batch(const BsedPackJob* jobs /*host*/, int njobs, void* stream);
int bsed_igemm3s(const BsedIgemmDesc* desc, int G, void* stream);
/* The same contraction as bsed_igemm3 (same reference layers: src/models/CNN.py:46-47, the seven nn.Conv2d; GRU input
 * projections src/models/RNN.py:7-16) in the round-3 structure (csrc/igemm3n.hip): a wave owns 32 output channels and
 * fetches their weight fragments straight from global memory (w = bsed_pack_weight3s table, K any multiple of 32), the
 * activation patch is double-buffered in LDS, one barrier per 32-channel chunk.  `out` is bit-identical to
 * bsed_igemm3's; STATS writes bsed_igemm3n_stats_rows(desc) partial rows of (2, N). */
int bsed_igemm3n(const BsedIgemmDesc* desc /*host*/, void* stream);
int bsed_igemm3n_stats_rows(const BsedIgemmDesc* desc /*host: NB, H, W, TH, TW, NP*/);
int bsed_igemm3n_variant(const BsedIgemmDesc* desc);   /* NWN | MW << 4 | PV << 8 | WPE << 12 of the build */
void bsed_igemm3n_set_wpe(int knob);  /* A/B knob: 2 / 3 = the BN = 128 build for that many waves per SIMD whatever the
                                       * shape, + 8 = no raised wave priority outside the MFMA loop, 0 = default */
void bsed_igemm3n_set_shape(int shape); /* A/B knob: MFMA shape of the nine-tap instances, 16 = v_mfma_f32_16x16x32_bf16 (default with
                                         * fp32 activations), 32 = v_mfma_f32_32x32x16_bf16 (default with bf16 activations, and
                                         * bit-identical to bsed_igemm3); 0 = defaults / BSED_IGEMM3N_SHAPE */
int bsed_igemm3s_auto_g(void);
int bsed_igemm3s_auto_g2(int CIN, int N);   /* per shape (resident workgroups differ with the LDS footprint) */

/* dW[tap][k][n] = sum_p in[p + (dh,dw)(tap)][k] * dy[p][n]: persistent workgroups over position tiles
 * write partial slabs part[G][ntaps][CINP][NP]; bsed_reduce_partials sums them into the gradient. */
typedef struct BsedWgradDesc {
  const float* in;       /* (NB,H,W,in_pitch) */
  const float* dy;       /* (NB,H,W,dy_pitch) */
  float* part;           /* (G, ntaps, CINP, NP) */
  const float* a_scale;  /* optional per-CIN affine on `in` */
  const float* a_shift;
  int in_pitch, dy_pitch;
  int NB, H, W, CIN, CINP, N, NP, G;
  int TH, TW, tilesH, tilesW, hh, hw;
  int ntaps, dh[9], dw[9];
  /* BatchNorm backward applied on load (bsed_wgrad3 only; all four NULL = plain dy).  With bn_y set, `dy` holds
   * g = dL/d(BatchNorm output) and the contraction runs on d_y = A g + B (bn_y - mean) + C, bn_coef = (3,N) [A|B|C] from
   * bsed_bn_bwd, bn_y / dy_out laid out like dy.  dy_out (optional) receives d_y, written once per element, for the
   * data-gradient convolution that follows: the separate apply pass over g and y (3 tensor passes) is gone. */
  const float* bn_y;
  const float* bn_coef;
  const float* bn_mean;
  float* dy_out;
  int act_bf16;          /* bsed_wgrad3: 1 = in, dy, bn_y and dy_out are bf16 tensors ("bf16" mode: one bf16 MFMA per product,
                          * the BatchNorm-backward map and the accumulation in fp32); 0 = fp32 */
} BsedWgradDesc;

int bsed_wgrad(const BsedWgradDesc* desc /*host*/, void* stream);
/* recommended number of partial slabs G for this shape (pointers in desc are ignored) */
int bsed_wgrad_auto_g(const BsedWgradDesc* desc /*host*/);
/* template instance bsed_wgrad launches for this shape, as MAXS*16 + NW (labels profiles and bench lines) */
int bsed_wgrad_variant(const BsedWgradDesc* desc /*host*/);
/* the same contraction with split-fp32 operands on the bf16 matrix cores (bf16x3, ~1e-5 relative): both tiles stay
 * position-major in LDS (bf16 hi / lo planes) and the operand fragments come out of transposing LDS reads; multi-tap
 * shapes whose two tile buffers fit in LDS run the producer/consumer kernel (wgrad3p_kernel).  Same descriptor, same
 * partial-slab layout; G must come from bsed_wgrad3_auto_g (it differs between the two kernels).
 * bsed_wgrad3_variant: MAXS*16 + NW, NW == 1 meaning wgrad3p_kernel<MAXS>; bit 12 = the BS template argument of
 * wgrad3_kernel<MAXS, NW, BS> (labels only).  BSED_WGRAD3_NOPIPE=1 in the environment forces wgrad3_kernel (A/B runs). */
int bsed_wgrad3(const BsedWgradDesc* desc /*host*/, void* stream);
int bsed_wgrad3_auto_g(const BsedWgradDesc* desc /*host*/);
int bsed_wgrad3_variant(const BsedWgradDesc* desc /*host*/);
/* dst[tap*s_tap + k*s_k + n*s_n] (+)= sum_g part[g][tap][k][n]   (k < K, n < N) */
int bsed_reduce_partials(const float* part, int G, int ntaps, int KP, int NP, int K, int N, float* dst,
                         long s_tap, long s_k, long s_n, int accumulate, void* stream);
/* several bsed_reduce_partials in one launch (the host queues the reductions whose results only the optimizer needs
 * and flushes them together): same arithmetic contract per job, jobs of one call must write distinct destinations */
#define BSED_REDUCE_MAX_JOBS 40
typedef struct BsedReduceJob {
  const float* part; float* dst;
  int G, ntaps, KP, NP, K, N, accumulate;
  long s_tap, s_k, s_n;
} BsedReduceJob;
int bsed_reduce_partials_batch(const BsedReduceJob* jobs /*host*/, int njobs, void* stream);
/* dst[tap][k][n] = src[tap*s_tap + k*s_k + n*s_n], zero for N <= n < NP */
int bsed_pack_weight(const float* src, float* dst, int ntaps, int K, int N, int NP, long s_tap, long s_k,
                     long s_n, void* stream);

/* ------------------------------------------------------------------------------------------------
 * Streaming CNN pieces (csrc/cnn_ops.hip)
 * ---------------------------------------------------------------------------------------------- */
/* first conv, Cin = 1 (src/models/CNN.py:46-47 with i = 0): x (NB,H,W) -> y (NB,H,W,CO); w is the
 * PyTorch (CO,1,3,3) tensor; stats (bsed_conv0_num_tiles, 2, CO) per-tile (sum, sumsq) or NULL. */
int bsed_conv0_fwd(const float* x, const float* w, const float* bias, float* y, float* stats, int NB, int H,
                   int W, int CO, void* stream);
int bsed_conv0_num_tiles(int NB, int H, int W);
/* dW of the first conv: part (G, 9, CO) partial slabs for bsed_reduce_partials.  With y / coef / mean (together or
 * all NULL) dy is the gradient w.r.t. the BatchNorm OUTPUT and BatchNorm's backward (coef from bsed_bn_bwd in
 * coefficients-only mode) is applied on load: the first block's d_y never touches HBM. */
int bsed_conv0_wgrad(const float* x, const float* dy, const float* y, const float* coef, const float* mean,
                     float* part, int G, int NB, int H, int W, int CO, void* stream);
/* fp64 scratch needed by the statistics reductions below (not to be shared by calls running concurrently on two
 * streams) */
size_t bsed_stats_scratch_bytes(int C);
/* BatchNorm2d(eps, momentum) in train mode (src/models/CNN.py:49): per-tile partials -> batch mean /
 * invstd, scale = gamma*invstd, shift = beta - mean*scale; running stats (unbiased var) and
 * num_batches_tracked are updated in place when given. count = NB*H*W. */
int bsed_bn_finalize(const float* partial, long ntiles, int C, double count, float eps, float momentum,
                     const float* gamma, const float* beta, float* running_mean, float* running_var,
                     long long* num_batches_tracked, float* mean, float* invstd, float* scale, float* shift,
                     void* scratch, void* stream);
/* eval mode: scale/shift from the running statistics */
int bsed_bn_eval(int C, float eps, const float* gamma, const float* beta, const float* running_mean,
                 const float* running_var, float* scale, float* shift, void* stream);
/* the same for every BatchNorm layer of an eval-mode forward in one launch */
#define BSED_BN_EVAL_MAX_JOBS 16
typedef struct BsedBnEvalJob {
  const float *gamma, *beta, *running_mean, *running_var;
  float *scale, *shift;
  int C;
} BsedBnEvalJob;
int bsed_bn_eval_batch(const BsedBnEvalJob* jobs /*host*/, int njobs, float eps, void* stream);
/* BatchNorm backward: partial = per-tile (sum g, sum g*y); writes dgamma/dbeta and turns g (n_elems,
 * NHWC, in place) into d_y = A g + B (y - mean) + C;  coef (3,C) receives [A | B | C].  g_inout = y = NULL:
 * coefficients only (the consumer applies the map on load). */
int bsed_bn_bwd(const float* partial, long ntiles, int C, double count, const float* gamma, const float* mean,
                const float* invstd, float* dgamma, float* dbeta, int accumulate, float* g_inout, const float* y,
                long n_elems, float* coef, void* scratch, void* stream);
/* dst[c] (+)= sum_tiles partial[tile][which][c] */
int bsed_stats_to_grad(const float* partial, long ntiles, int C, int which, float* dst, int accumulate,
                       void* scratch, void* stream);
/* dst[c] (+)= sum_r in[r*pitch + c], r < M; part holds G*2*C floats */
int bsed_colsum(const float* in, long M, int C, int pitch, float* part, int G, float* dst, int accumulate,
                void* scratch, void* stream);
/* first stage of bsed_colsum alone: part (G, 2, C), G <= M, slot 0 = per-workgroup column sums (second stage: a
 * bsed_reduce_partials_batch job with KP = 2, K = 1) */
int bsed_colsum_part(const float* in, long M, int C, int pitch, float* part, int G, void* stream);
/* nn.Dropout(p) with a stateless Philox mask: out = in * keep/(1-p); the same call is its backward */
int bsed_dropout(const float* in, float* out, long n, float p, uint32_t rng_stream, uint64_t seed, void* stream);

/* GLU stage of a 16-channel block as HBM-bound streaming kernels (csrc/glu_small.hip): same math as
 * bsed_igemm(BSED_EPI_GLU_POOL) / the GLU backward chain (src/models/CNN.py:5-16,59-67), one pass over y.
 *   forward : y (B,H,W,16) -> pooled (B,H/ph,W/pw,16)
 *   backward: y, dpool -> g = dL/d(BN output) (B,H,W,16), per-workgroup partials part_dw (G,16,16),
 *             part_db (G,2,16) [slot 0], part_st (G,2,16) = (sum g, sum g*y) for bsed_bn_bwd */
int bsed_glu16_fwd(const float* y, const float* scale, const float* shift, const float* wg, const float* bg,
                   float* out, int B, int H, int W, int C, int ph, int pw, float drop_p, uint32_t rng_stream,
                   uint64_t seed, void* stream);
int bsed_glu16_bwd(const float* y, const float* scale, const float* shift, const float* wg, const float* bg,
                   const float* dpool, float* g_out, float* part_dw, float* part_db, float* part_st, int G, int B,
                   int H, int W, int C, int ph, int pw, float drop_p, uint32_t rng_stream, uint64_t seed,
                   void* stream);

/* The whole first CNN block (conv3x3 1->16, BatchNorm, GLU, dropout, avg-pool: src/models/CNN.py:46-67, i = 0) WITHOUT
 * its conv output and gradient tensors in HBM (csrc/block0.hip): y0 is a 9-tap stencil of the input and is recomputed
 * where it is needed; BatchNorm's batch statistics come from x alone; conv0's weight gradient is assembled from
 * Gx = sum g x_tap (taken by the backward kernel) and the input's tap correlations R / Sx (taken with the statistics).
 *   stats : x (NB,H,W) -> stats (G,2,16) per-workgroup (sum y, sum y^2) for bsed_bn_finalize; xr_part (G,54) scratch;
 *           xr64 (54) fp64 totals [Sx (9) | R upper triangle (45)] for bsed_block0_wgrad_finish
 *   fwd   : x -> pooled (B,H/ph,W/pw,16), same values (bit for bit) as bsed_conv0_fwd + bsed_glu16_fwd
 *   bwd   : x, dpool -> per-workgroup partials part_dw (G,16,16), part_db (G,2,16) [slot 0], part_st (G,2,16) =
 *           (sum g, sum g*y) for bsed_bn_bwd (coefficients only), part_gx (G,9,16)
 *   wgrad_finish : dst (16,1,3,3) (+)= A Gx + B (Yx - mean Sx) + C Sx with coef = [A|B|C] of bsed_bn_bwd */
int bsed_block0_stats(const float* x, const float* cw_t /* the (16,1,3,3) weight transposed: [tap][channel] */,
                      const float* cb, float* stats, float* xr_part, double* xr64,
                      int G, int NB, int H, int W, int CO, void* stream);
int bsed_block0_fwd(const float* x, const float* cw, const float* cb, const float* scale, const float* shift,
                    const float* wg, const float* bg, float* out, int B, int H, int W, int CO, int ph, int pw,
                    float drop_p, uint32_t rng_stream, uint64_t seed,
                    int act_bf16 /* 1: the pooled output / its gradient is a bf16 tensor (bf16 mode) */, void* stream);
int bsed_block0_bwd(const float* x, const float* cw, const float* cb, const float* scale, const float* shift,
                    const float* wg, const float* bg, const float* dpool, float* part_dw, float* part_db,
                    float* part_st, float* part_gx, int G, int B, int H, int W, int CO, int ph, int pw, float drop_p,
                    uint32_t rng_stream, uint64_t seed,
                    int act_bf16 /* 1: the pooled output / its gradient is a bf16 tensor (bf16 mode) */, void* stream);
int bsed_block0_wgrad_finish(const float* part_gx, int G, const double* xr64, const float* coef, const float* mean,
                             const float* cw, const float* cb, float* dst, int accumulate, int CO, void* stream);

/* Fused GLU backward for C in {32,64,128} (csrc/glu_bwd.hip): y, d_pooled -> g = dL/d(BN output) in one pass, with
 * the three contractions (lin recompute, g, dW) chained on chip.  wfwd = bsed_pack_weight'ed W^T ([c][n]),
 * wbwd = the (C,C) Linear weight itself.  Outputs per-workgroup partials: part_dw (G*bsed_glu_bwd_slabs(C), C, C),
 * part_db (G,2,C) [slot 0 = sum d_lin], part_st (G,2,C) = (sum g, sum g*y) for bsed_bn_bwd. */
int bsed_glu_bwd_fused(const float* y, const float* scale, const float* shift, const float* wfwd,
                       const float* wbwd, const float* bias, const float* dpool, float* g, float* part_dw,
                       float* part_db, float* part_st, int G, int NB, int H, int W, int C, int TH, int TW, int ph,
                       int pw, float drop_p, uint32_t rng_stream, uint64_t seed, void* stream);
int bsed_glu_bwd_slabs(int C);

/* The same fused GLU backward in the split-fp32 ("bf16x3") contraction mode for C in {32,64} (csrc/glu3.hip): all
 * three contractions on the bf16 matrix cores, activations never staged through LDS.  w = the (C,C) Linear weight.
 * part_dw holds G * bsed_glu_bwd3_slabs(C) slabs; bsed_glu_bwd3_auto_g(C) = workgroups of one resident round. */
int bsed_glu_bwd3(const float* y, const float* scale, const float* shift, const float* w, const float* bias,
                  const float* dpool, float* g, float* part_dw, float* part_db, float* part_st, int G, int NB, int H,
                  int W, int C, int TH, int TW, int ph, int pw, float drop_p, uint32_t rng_stream, uint64_t seed,
                  int act_bf16 /* 1: y, d_pooled, g, d_lin, pooled are bf16 tensors (bf16 mode) */, void* stream);
int bsed_glu_bwd3_slabs(int C);
int bsed_glu_bwd3_auto_g(int C);
/* C = 128 in the split-fp32 mode: g, db and BatchNorm partials as above, but d_lin (NB,H,W,128) is written out and
 * the Linear weight gradient is a separate 1-tap bsed_wgrad3(in = y with a_scale/a_shift, dy = d_lin).  frag_table:
 * bsed_glu_bwd3n_table_bytes() of device scratch (filled by the call: both weight operands pre-split). */
int bsed_glu_bwd3n(const float* y, const float* scale, const float* shift, const float* w, const float* bias,
                   const float* dpool, float* g, float* dlin, float* part_db, float* part_st, void* frag_table, int G,
                   int NB, int H, int W, int C, int TH, int TW, int ph, int pw, float drop_p, uint32_t rng_stream,
                   uint64_t seed,
                  int act_bf16 /* 1: y, d_pooled, g, d_lin, pooled are bf16 tensors (bf16 mode) */, void* stream);
size_t bsed_glu_bwd3n_table_bytes(void);
/* Forward of the same stage in the split-fp32 mode, C in {32,64,128}: y (NB,H,W,C) -> pooled (NB,H/ph,W/pw,C);
 * replaces bsed_igemm(BSED_EPI_GLU_POOL) (GLU.forward + nn.Dropout + nn.AvgPool2d, src/models/CNN.py:5-16,59-67).
 * Vertical pooling (ph = 2) is supported for tile widths TW in {2,8,16}. */
int bsed_glu_fwd3(const float* y, const float* scale, const float* shift, const float* w, const float* bias,
                  float* pooled, int G, int NB, int H, int W, int C, int TH, int TW, int ph, int pw, float drop_p,
                  uint32_t rng_stream, uint64_t seed,
                  int act_bf16 /* 1: y, d_pooled, g, d_lin, pooled are bf16 tensors (bf16 mode) */, void* stream);
int bsed_glu_fwd3_auto_g(int C);

/* ------------------------------------------------------------------------------------------------
 * Clip-level domain discriminator glue (csrc/disc.hip); replaces Clip_Discriminator.forward
 * (src/models/CRNN_GRL.py:16-53), GradientReverseFunction (src/DA/grl.py:12-22) and the BCE of
 * ConditionalDomainAdversarialLoss.forward (src/DA/cdan_frame.py:89-119).  The 3x3 / stride-2 / pad-0
 * convolutions run as im2col + bsed_igemm / bsed_wgrad; images are (N, H = time, W = feature, C).
 * ---------------------------------------------------------------------------------------------- */
/* Frame-level discriminator glue (csrc/disc.hip; reference Frame_Discriminator, src/models/CRNN_GRL.py:116-140).
 *   bsed_leaky_dropout_fwd/bwd: out = LeakyReLU_slope(a) * dropout mask; d_a = d_out * mask * LeakyReLU'(a) (n % 4 == 0;
 *     the mask is a counter hash of (seed, rng_stream, element index), regenerated in backward);
 *   bsed_frame_head_fwd: d (M) = sigmoid(x (M,32) . w (32) + b);
 *   bsed_frame_head_bwd: dx (M,32) and part (G,2,32): row 0 = partial dw, row 1 column 0 = partial db (sum over G). */
int bsed_leaky_dropout_fwd(const float* a, float* out, long n, float slope, float drop_p, uint32_t rng_stream,
                           uint64_t seed, void* stream);
int bsed_leaky_dropout_bwd(const float* d_out, const float* a, float* d_a, long n, float slope, float drop_p,
                           uint32_t rng_stream, uint64_t seed, void* stream);
int bsed_frame_head_fwd(const float* x, const float* w, const float* b, float* d, long M, int K, void* stream);
int bsed_frame_head_bwd(const float* x, const float* w, const float* d, const float* d_out, float* dx, float* part, int G,
                        long M, int K, void* stream);

/* Space-to-depth glue of the direct stride-2 convolutions (csrc/disc.hip): a 3x3 / stride-2 / pad-0 convolution over
 * A (Hi,Wi,C) equals a 2x2 / stride-1 convolution over X'[p][q][(a*2+b)*C + c] = A[2p+a][2q+b][c] (4C channels, taps
 * (dp,dq) in {0,1}^2, weight slot (dp,dq,a,b) = W[2dp+a][2dq+b] or zero), which the implicit-GEMM kernels run without
 * an im2col matrix.
 *   bsed_s2d_fwd: X' (N,Hp,Wp,4C), Hp = ceil(Hi/2), Wp = ceil(Wi/2), from y (N,Ha,Wa,C) (allocated extent >= valid
 *                 extent Hi x Wi) with BatchNorm apply + LeakyReLU(0.2) fused (scale/shift NULL: identity); pixels
 *                 beyond Hi x Wi are zero.
 *   bsed_s2d_bwd: g (N,Ha,Wa,C) = dX' gathered back * LeakyReLU'(y*scale+shift) on the valid extent, zero elsewhere, and
 *                 per-block partial sums (sum g, sum g*y) for the BatchNorm backward: stats (bsed_s2d_num_blocks, 2, C). */
int bsed_s2d_fwd(const float* y, const float* scale, const float* shift, float* xp, int N, int Ha, int Wa, int Hi, int Wi,
                 int C, void* stream);
int bsed_s2d_num_blocks(int N, int Ha, int Wa, int C);
int bsed_s2d_bwd(const float* dxp, const float* y, const float* scale, const float* shift, float* g, float* stats, int N,
                 int Ha, int Wa, int Hi, int Wi, int C, void* stream);

/* col[(n,ho,wo)][(dw*3+dh)*CP + c] = leaky_relu_0.2(act*scale+shift) (identity when scale == NULL);
 * C == 1 writes 16 columns (9 taps + zeros).  Ho = (Hi-3)/2+1, Wo = (Wi-3)/2+1. */
int bsed_im2col_s2(const float* act, const float* scale, const float* shift, float* col, int N, int Hi, int Wi,
                   int C, int CP, void* stream);
/* adjoint gather.  C > 1: out = d_act * leaky'(y*scale+shift), stats (bsed_col2im_s2_num_blocks, 2, C) =
 * per-block (sum g, sum g*y) for bsed_bn_bwd.  C == 1: out = d_act * out_scale (GRL: out_scale = -lambda). */
int bsed_col2im_s2(const float* dcol, const float* y, const float* scale, const float* shift, float* out,
                   float* stats, int N, int Hi, int Wi, int C, int CP, float out_scale, void* stream);
int bsed_col2im_s2_num_blocks(int N, int Hi, int Wi, int C);
/* BN5 + LeakyReLU + AdaptiveAvgPool2d((2,1)) + Linear(16,1) + sigmoid (+ BCE vs label 1 for n < Ns else 0,
 * mean over N, and the backward to g5 = dL/d(BN5 output) with per-sample partials) */
int bsed_disc_head(const float* y5, const float* scale, const float* shift, const float* wl, const float* bl, int N,
                   int Ns, int H5, int W5, int C5, int train, float* d_out, float* g5, float* stats, float* dwl_part,
                   float* dbl_part, float* loss_part, void* stream);

/* ------------------------------------------------------------------------------------------------
 * Feature-pyramid glue of CRNN_fpn (csrc/fpn.hip); replaces nn.Upsample((T_out,1), mode='bilinear',
 * align_corners=True) on width-1 maps (src/models/CRNN_GRL.py:333-336,378-384) and its backward.
 * Tensors are (B, T, C) with a row pitch, so the output can be the right half of a concatenation buffer.
 * ---------------------------------------------------------------------------------------------- */
int bsed_upsample_time_fwd(const float* in, float* out, int B, int T_in, int T_out, int C, int in_pitch, int out_pitch,
                           void* stream);
int bsed_upsample_time_bwd(const float* dout, float* din, int B, int T_in, int T_out, int C, int dout_pitch,
                           int din_pitch, void* stream);

/* ------------------------------------------------------------------------------------------------
 * Bidirectional GRU recurrence (csrc/gru.hip); replaces nn.GRU of src/models/RNN.py:7-16.
 *   xp    (B,T,768)  x @ [W_ih; W_ih_reverse]^T + b_ih   (from bsed_igemm), [dir*384 + gate*128 + k]
 *   w_hh  (2,384,128), b_hh (2,384)                       gate order r,z,n
 *   out   (B,T,256)  [dir*128 + k];  gates (B,T,2,4,128) r,z,n,(W_hn h + b_hn) saved for backward
 * ---------------------------------------------------------------------------------------------- */
int bsed_gru_fwd(const float* xp, const float* w_hh, const float* b_hh, float* out, float* gates, int B, int T,
                 int rows_per_wg, void* stream);
/* BPTT: dxp = d(x-side pre-activations) (B,T,768), dgh = d(h-side pre-activations) (B,T,768) */
int bsed_gru_bwd(const float* dout, const float* out, const float* gates, const float* w_hh, float* dxp,
                 float* dgh, int B, int T, int rows_per_wg, void* stream);
/* The same recurrence with h @ W_hh^T of every step on the bf16 matrix cores with split-fp32 operands (4 batch rows
 * per workgroup; the default in the bf16x3 contraction mode).  Same tensors, same layouts, same reference lines. */
int bsed_gru_fwd3(const float* xp, const float* w_hh, const float* b_hh, float* out, float* gates, int B, int T,
                  void* stream);
/* part_bih / part_bhh (nullable, together): (bsed_gru_bwd3_rows(B), 768) per-batch-row partial sums over time of
 * dxp / dgh, i.e. the bias gradients before the final column sum (bsed_colsum over those few rows). */
int bsed_gru_bwd3(const float* dout, const float* out, const float* gates, const float* w_hh, float* dxp,
                  float* dgh, float* part_bih, float* part_bhh, int B, int T, void* stream);
int bsed_gru_bwd3_rows(int B);

/* ------------------------------------------------------------------------------------------------
 * Predictor head + losses (csrc/head.hip); replaces Predictor.forward (src/models/CRNN_GRL.py:441-460)
 * and the BCE / MSE loss assembly of train_mt (src/main_baseline.py:431-498).
 *   w (2C,K): rows 0..C-1 = dense.weight, rows C..2C-1 = dense_softmax.weight; b (2C) likewise.
 * ---------------------------------------------------------------------------------------------- */
/* out (B,C) = max over time of y (B,T,C): the clip-level targets the train loop derives from the strong ones
 * (src/main_baseline.py train_mt: target_weak = target.max(-2)[0]) */
int bsed_max_over_time(const float* y, float* out, int B, int T, int C, void* stream);
/* The head kernels split each clip's frames over S = bsed_head_splits(B, T) workgroups.  bsed_head_fwd needs a
 * (B,S,2,C) scratch buffer `part` when S > 1; bsed_head_bwd writes S rows per clip into its partial outputs
 * (dw_part (B*S,2C,K), db_part (B*S,2C), loss_part (B*S,6)): sum over the leading dimension as before. */
int bsed_head_splits(int B, int T);
int bsed_head_fwd(const float* x, const float* w, const float* b, float* strong, float* sof_raw, float* weak,
                  float* den, float* part, int B, int T, int K, int C, int attention, void* stream);

/* CNN-only tagging head (csrc/tag.hip); replaces the tail of CRNN_pred.forward (src/models/CRNN_GRL.py:252-290):
 * strong = sigmoid(x), sof = clamp(softmax_class(logits), 1e-7, 1), weak = sum_t strong*sof / sum_t sof.
 *   x, logits, strong (B,T,C); weak (B,C); part: (B, bsed_tag_splits(B,T), 2, C) scratch.  C = 128. */
int bsed_tag_splits(int B, int T);
int bsed_tag_head_fwd(const float* x, const float* logits, float* strong, float* weak, float* part, int B, int T, int C,
                      void* stream);

/* get_predictions post-processing (src/evaluation_measures.py:188-205): out = median_filter(strong > threshold,
 * size=(win,1)) with scipy.ndimage's window origin and 'reflect' boundary; (B,T,C) float 0/1 mask */
int bsed_binarize_median(const float* strong, float* out, int B, int T, int C, float threshold, int win,
                         void* stream);

/* Contiguous-region decode + frames -> seconds on the GPU (ManyHotEncoder.decode_strong, src/utilities/ManyHotEncoder.py:
 * 148-164, and src/evaluation_measures.py:205-209).  mask (B,T,C) 0/1 from bsed_binarize_median.
 *   bsed_decode_count: counts (B*C) = number of on-runs per (clip, class) column;
 *   bsed_decode_write: offsets (B*C) = exclusive prefix sum of counts (caller's), E = total; writes ev_clip (E),
 *     ev_class (E), ev_frames (E,2) [onset, offset) and ev_seconds (E,2) = clip(frame * scale, 0, max_len) in float64,
 *     ordered by clip, class, time -- the reference's row order. */
int bsed_decode_count(const float* mask, int B, int T, int C, int* counts, void* stream);
int bsed_decode_write(const float* mask, const int* offsets, int B, int T, int C, double scale, double max_len,
                      int* ev_clip, int* ev_class, int* ev_frames, double* ev_seconds, void* stream);

typedef struct BsedHeadBwdDesc {
  const float* x;            /* (B,T,K) encoder output */
  const float* w;            /* (2C,K) */
  const float* strong;       /* (B,T,C) saved by bsed_head_fwd */
  const float* sof_raw;      /* (B,T,C) unclamped softmax */
  const float* weak;         /* (B,C) */
  const float* den;          /* (B,C) sum_t clamp(softmax) */
  const float* y_strong;     /* (B,T,C) or NULL: BCE(strong, y) * w_strong */
  const float* y_weak;       /* (B,C)   or NULL: BCE(weak, y)   * w_weak */
  const float* ema_strong;   /* (B,T,C) or NULL: MSE(strong, ema) * w_cons_s */
  const float* ema_weak;     /* (B,C)   or NULL: MSE(weak, ema)   * w_cons_w */
  const float* ema_strong2;  /* (B,T,C) or NULL: second MSE target on strong * w_cons_s2 (ISP shift consistency) */
  const float* g_strong_ext; /* (B,T,C) or NULL: upstream dL/dstrong (autograd drop-in path) */
  const float* g_weak_ext;   /* (B,C)   or NULL */
  float w_strong, w_weak, w_cons_s, w_cons_w, w_cons_s2;
  float inv_n_strong, inv_n_weak; /* 1/(B*T*C), 1/(B*C): reduction='mean' */
  float* dx;                 /* (B,T,K) */
  float* dw_part;            /* (B,2C,K) per-clip partials for bsed_reduce_partials */
  float* db_part;            /* (B,2C) */
  float* loss_part;          /* (B,6): plain sums of BCE_strong, BCE_weak, SE_strong, SE_weak, SE_strong2, 0 */
  int B, T, K, C, attention;
} BsedHeadBwdDesc;

int bsed_head_bwd(const BsedHeadBwdDesc* desc /*host*/, void* stream);

/* ------------------------------------------------------------------------------------------------
 * Flat-arena optimizer / EMA (csrc/optim.hip)
 * ---------------------------------------------------------------------------------------------- */
/* torch.optim.Adam step (src/main_baseline.py:861-867); grad_scale multiplies g first (1/world_size) */
int bsed_adam_step(float* p, const float* g, float* m, float* v, long n, float lr, float beta1, float beta2,
                   float eps, float weight_decay, long step, float grad_scale, void* stream);
/* torch.optim.SGD(momentum, nesterov, weight_decay) step (src/main_scmt_ada_weak.py:854-866) */
int bsed_sgd_step(float* p, const float* g, float* buf, long n, float lr, float momentum, float weight_decay,
                  int first_step, int nesterov, float grad_scale, void* stream);
/* update_ema_variables (src/main_baseline.py:91-105): ema = ema*alpha + p*(1-alpha) */
int bsed_ema_update(float* ema, const float* p, long n, float alpha, void* stream);
int bsed_ema_update_i64(long long* ema, const long long* p, int n, float alpha, void* stream);
int bsed_scale(float* x, long n, float s, void* stream);
/* per-sample circular shift (torch.roll) of a (B,H,W) tensor: out[b][h][w] = in[b][(h-sh[b]) mod H][(w-sw[b]) mod W];
 * sh / sw are DEVICE int32 arrays (B) or NULL.  Reference src/main_baseline.py:229-277,374-401 (ISP views). */
int bsed_roll(const float* in, float* out, int B, int H, int W, const int* sh, const int* sw, void* stream);
/* y += a * x  (sums the class-loss and domain-loss gradients at the encoder output) */
int bsed_axpy(float* y, const float* x, long n, float a, void* stream);

/* ------------------------------------------------------------------------------------------------
 * Self tests (used by tests/ only)
 * ---------------------------------------------------------------------------------------------- */
/* C(32,32) = A(32,K) @ B(K,32) through one wave of v_mfma_f32_32x32x2_f32: pins the fragment maps */
int bsed_selftest_mfma(const float* A, const float* B, float* C, int K, void* stream);
/* the same through v_mfma_f32_32x32x16_bf16 with bf16x3 split operands (K a multiple of 16) */
int bsed_selftest_mfma_bf16x3(const float* A, const float* B, float* C, int K, void* stream);

/* Device-resident step state for HIP-graph replays of a train step (csrc/capi.hip): when set, every kernel that takes a
 * dropout seed adds *seed_add_dev (uint64) to it and the Adam kernel adds *step_add_dev (int) to its step count;
 * bsed_step_state_advance is the graph node that bumps both after a step.  NULL, NULL = eager mode (the default). */
int bsed_set_step_state(const void* seed_add_dev, const void* step_add_dev);
int bsed_step_state_advance(void* seed_add_dev, void* step_add_dev, uint64_t seed_inc, int step_inc, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* BSED_H */
