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
int bsed_mel_num_frames(const void* plan, int n_samples);   /* 1 + n_samples / hop */
/* wav (B, n_samples) -> LINEAR mel amplitude (B, T, n_mels) == preprocess(audio).  Also emits the
 * per-clip max (B) and the per-(clip, band) sum over time of x^2 (B, n_mels), which the dB clamp and
 * the SNR noise need, so neither costs another pass over HBM. */
int bsed_mel_linear(const void* plan, const float* wav, int B, int n_samples, float* mel_lin,
                    float* clip_max, float* bin_sumsq, void* stream);
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
 * Self tests (used by tests/ only)
 * ---------------------------------------------------------------------------------------------- */
/* C(32,32) = A(32,K) @ B(K,32) through one wave of v_mfma_f32_32x32x2_f32: pins the fragment maps */
int bsed_selftest_mfma(const float* A, const float* B, float* C, int K, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* BSED_H */
