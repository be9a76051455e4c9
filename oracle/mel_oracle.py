"""Mel-stage oracle (numpy).  TEST INFRASTRUCTURE ONLY -- never on the product path.

**parity unpinned**: the arithmetic of this stage lives in librosa (unpinned,
un-vendored third-party dependency of the reference, not installed here).  This
file restates what the reference's call sites ask librosa to do:

  * ``preprocess``            <- /root/reference/src/data/preprocess.py:18-45
                                 (= src/synth_data/synth_data_preprocess.py:15-42)
  * ``amplitude_to_db``       <- src/data/Transforms.py:74-86 (ApplyLog ->
                                 librosa.amplitude_to_db(S, ref=1, amin=1e-5, top_db=80))
  * ``gaussian_noise``        <- src/data/Transforms.py:155-178
  * ``pad_trunc_seq``         <- src/data/Transforms.py:89-109
  * ``transform_pair``        <- src/data/Transforms.py:304-322 (noise -> log -> pad -> tensor)

librosa semantics restated (documented behaviour): stft(center=True, reflect pad
n_fft//2, frames at hop, window multiply in float64, rfft, result stored complex64);
filters.mel(htk=False, norm=None) Slaney scale, float32 basis; melspectrogram(S=...)
= basis @ S in float32; amplitude_to_db = power_to_db(S**2, amin**2) with the
top_db clamp against the per-array max.

Second source (not a pin): tests/test_mel_oracle_crosscheck_cpu.py compares this file with
transformers.audio_utils (a numpy port of librosa) and scipy.signal.stft / torch.stft at
both configurations of the path.
"""
import numpy as np

SR = 32000
N_FFT = 2048
HOP = 255
N_MELS = 128
FMIN = 0.0
FMAX = 16000.0


# ----------------------------------------------------------------------------- mel scale
def hz_to_mel(f):
    f = np.asanyarray(f, dtype=np.float64)
    f_sp = 200.0 / 3
    mels = f / f_sp
    min_log_hz = 1000.0
    min_log_mel = min_log_hz / f_sp
    logstep = np.log(6.4) / 27.0
    if f.ndim:
        m = f >= min_log_hz
        mels[m] = min_log_mel + np.log(f[m] / min_log_hz) / logstep
    elif f >= min_log_hz:
        mels = min_log_mel + np.log(f / min_log_hz) / logstep
    return mels


def mel_to_hz(m):
    m = np.asanyarray(m, dtype=np.float64)
    f_sp = 200.0 / 3
    freqs = f_sp * m
    min_log_hz = 1000.0
    min_log_mel = min_log_hz / f_sp
    logstep = np.log(6.4) / 27.0
    if m.ndim:
        lg = m >= min_log_mel
        freqs[lg] = min_log_hz * np.exp(logstep * (m[lg] - min_log_mel))
    elif m >= min_log_mel:
        freqs = min_log_hz * np.exp(logstep * (m - min_log_mel))
    return freqs


def mel_frequencies(n_mels, fmin, fmax):
    return mel_to_hz(np.linspace(hz_to_mel(fmin), hz_to_mel(fmax), n_mels))


def mel_filterbank(sr=SR, n_fft=N_FFT, n_mels=N_MELS, fmin=FMIN, fmax=FMAX):
    """(n_mels, 1+n_fft//2) float32 triangular Slaney filterbank, no area norm."""
    n_bins = 1 + n_fft // 2
    fftfreqs = np.linspace(0.0, sr / 2.0, n_bins)
    mel_f = mel_frequencies(n_mels + 2, fmin, fmax)
    fdiff = np.diff(mel_f)
    ramps = np.subtract.outer(mel_f, fftfreqs)
    w = np.zeros((n_mels, n_bins), dtype=np.float32)
    for i in range(n_mels):
        lower = -ramps[i] / fdiff[i]
        upper = ramps[i + 2] / fdiff[i + 1]
        w[i] = np.maximum(0, np.minimum(lower, upper))
    return w


# ----------------------------------------------------------------------------- STFT
def n_frames_for(n_samples, hop=HOP):
    return 1 + n_samples // hop


def stft_mag(audio, n_fft=N_FFT, hop=HOP):
    """|STFT| as float32 (n_bins, T): hamming(2048) symmetric, reflect-centre pad."""
    audio = np.asarray(audio, dtype=np.float32)
    win = np.hamming(n_fft)  # float64, symmetric (preprocess.py:19)
    ypad = np.pad(audio, n_fft // 2, mode="reflect")
    T = 1 + (len(ypad) - n_fft) // hop
    idx = hop * np.arange(T)[:, None] + np.arange(n_fft)[None, :]
    frames = ypad[idx]  # (T, n_fft) float32
    spec = np.fft.rfft(win[None, :] * frames, axis=-1)  # float64 math
    spec = spec.astype(np.complex64)  # librosa stores complex64 for float32 input
    return np.abs(spec).T  # float32 (n_bins, T)


def preprocess(audio, sr=SR, n_fft=N_FFT, hop=HOP, n_mels=N_MELS, fmin=FMIN, fmax=FMAX,
               compute_log=False):
    """(T, n_mels) float32 LINEAR mel amplitude, as preprocess.py:18-45."""
    S = stft_mag(audio, n_fft, hop)
    basis = mel_filterbank(sr, n_fft, n_mels, fmin, fmax)
    mel = (basis @ S).astype(np.float32)  # float32 matmul
    if compute_log:
        mel = amplitude_to_db(mel)
    return np.ascontiguousarray(mel.T).astype(np.float32)


# ----------------------------------------------------------------------------- dB
def amplitude_to_db(S, amin=1e-5, top_db=80.0):
    """librosa.amplitude_to_db(S) with ref=1: 10*log10(max(amin^2, S^2)), clamp max-80.
    dtype of S is preserved (float32 clean view, float64 noisy view)."""
    S = np.asarray(S)
    power = np.square(np.abs(S))
    log_spec = 10.0 * np.log10(np.maximum(amin * amin, power))
    log_spec = log_spec - 10.0 * np.log10(np.maximum(amin * amin, 1.0))
    return np.maximum(log_spec, log_spec.max() - top_db)


def gaussian_noise_std(features, snr=30.0):
    """Per-mel-bin std of the SNR noise (Transforms.py:173)."""
    return np.sqrt(np.mean((features ** 2) * (10 ** (-snr / 10)), axis=-2))


def gaussian_noise(features, snr=30.0, unit_noise=None, rng=None):
    """features + N(0, std_f).  ``unit_noise`` (same shape, N(0,1)) may be injected so
    that the GPU path can be checked on identical samples."""
    std = gaussian_noise_std(features, snr)
    if unit_noise is None:
        rng = rng or np.random
        unit_noise = rng.normal(0, 1, features.shape)
    return features + np.asarray(unit_noise, dtype=np.float64) * std.astype(np.float64)


def pad_trunc_seq(x, max_len):
    if x.shape[-2] <= max_len:
        pad = ((0, 0),) * (x.ndim - 2) + ((0, max_len - x.shape[-2]), (0, 0))
        return np.pad(x, pad, mode="constant")
    return x[..., :max_len, :]


def transform_pair(mel_lin, max_frames, snr=30.0, unit_noise=None):
    """get_transforms(): (clean_db, noisy_db), each float32 (1, max_frames, n_mels)."""
    noisy = gaussian_noise(mel_lin, snr, unit_noise=unit_noise)
    out = []
    for v in (mel_lin, noisy):
        db = amplitude_to_db(v.T).T
        db = pad_trunc_seq(db, max_frames)
        out.append(db.astype(np.float32)[None])
    return out[0], out[1]


# ----------------------------------------------------------------------------- synthetic clips
def synth_clip(index, sr=SR, seconds=10.0):
    """Deterministic synthetic clip of SURVEY.md section 8(d): 0.1*N(0,1) floor + 3
    tones/chirps; returns (float32 wave in [-1,1], [(onset_s, offset_s, class)])."""
    rng = np.random.default_rng(2023 + index)
    n = int(seconds * sr)
    t = np.arange(n, dtype=np.float64) / sr
    y = 0.1 * rng.standard_normal(n)
    events = []
    for _ in range(3):
        f0 = rng.uniform(500, sr / 2 - 500)
        f1 = rng.uniform(500, sr / 2 - 500) if rng.random() < 0.5 else f0
        amp = rng.uniform(0.05, 0.5)
        on = rng.uniform(0, seconds - 0.2)
        off = rng.uniform(on + 0.2, seconds)
        cls = int(rng.integers(0, 20))
        seg = (t >= on) & (t < off)
        tt = t[seg] - on
        dur = off - on
        phase = 2 * np.pi * (f0 * tt + 0.5 * (f1 - f0) * tt * tt / dur)
        y[seg] += amp * np.sin(phase)
        events.append((on, off, cls))
    y = np.clip(y, -1.0, 1.0).astype(np.float32)
    return y, events
