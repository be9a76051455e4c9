"""Mel front end on the GPU -- host-side mirror of the reference's feature code.

  preprocess(audio, compute_log=False)   <- reference src/data/preprocess.py:18-45
  MelFrontEnd.transform(...)             <- get_transforms(): AugmentGaussianNoise -> ApplyLog ->
                                            PadOrTrunc -> ToTensor (src/data/Transforms.py:304-322)

The reference runs these per clip on the CPU (librosa, offline + DataLoader).  Here a whole batch of
raw waveforms goes through three HIP kernels (csrc/mel.hip) and comes out as the (B,1,T,128) CRNN
input without leaving HBM.
"""
import ctypes
import math

import torch

from . import _lib as L
from . import ops


class MelConfig:
    """Constants of reference src/data/config.py:47-57 (``R`` config) as defaults."""

    def __init__(self, sr=32000, n_window=2048, hop_size=255, n_mels=128, mel_f_min=0.0, mel_f_max=None,
                 max_len_seconds=10.0, noise_snr=30.0, top_db=80.0):
        self.sr, self.n_window, self.hop_size, self.n_mels = sr, n_window, hop_size, n_mels
        self.mel_f_min = mel_f_min
        self.mel_f_max = min(16000.0, sr / 2.0) if mel_f_max is None else mel_f_max
        self.max_len_seconds, self.noise_snr, self.top_db = max_len_seconds, noise_snr, top_db

    @property
    def max_frames(self):
        return math.ceil(self.max_len_seconds * self.sr / self.hop_size)


class MelFrontEnd:
    def __init__(self, cfg=None):
        L._require_gpu()
        self.cfg = cfg or MelConfig()
        c = L.MelCfg(self.cfg.sr, self.cfg.n_window, self.cfg.hop_size, self.cfg.n_mels,
                     self.cfg.mel_f_min, self.cfg.mel_f_max)
        self._plan = ctypes.c_void_p()
        L.call("bsed_mel_plan_create", ctypes.byref(c), ctypes.byref(self._plan))
        self.nnz = int(L.lib().bsed_mel_plan_nnz(self._plan))
        self.frames_per_wave = int(L.lib().bsed_mel_plan_frames_per_wave(self._plan))   # 2: stft_mel2_kernel

    def __del__(self):
        try:
            if getattr(self, "_plan", None):
                L.lib().bsed_mel_plan_destroy(self._plan)
                self._plan = None
        except Exception:
            pass

    def num_frames(self, n_samples):
        return 1 + n_samples // self.cfg.hop_size

    def linear(self, wav):
        """(B, n) float32 GPU waveforms -> (mel_lin (B,T,n_mels), clip_max (B,), bin_sumsq (B,n_mels))."""
        if wav.dim() == 1:
            wav = wav[None]
        B, n = wav.shape
        T = self.num_frames(n)
        mel = torch.empty((B, T, self.cfg.n_mels), device=wav.device, dtype=torch.float32)
        cmax = torch.empty((B,), device=wav.device, dtype=torch.float32)
        sumsq = torch.empty((B, self.cfg.n_mels), device=wav.device, dtype=torch.float32)
        # algorithmic work: wave in + linear mel out; per frame a 2048-point real FFT (2.5 N log2 N), 1025
        # magnitudes and the sparse filterbank
        ops._note("stft_mel2_kernel" if self.frames_per_wave == 2 else "stft_mel_kernel", f"T{T}",
                  B * T * (2.5 * 2048 * 11 + 4.0 * 1025 + 2.0 * self.nnz),
                  4.0 * B * (n + T * self.cfg.n_mels))
        fn = L.lib().bsed_mel_scratch_floats
        fn.restype = ctypes.c_long
        scratch = torch.empty(fn(self._plan, L.c_int(B), L.c_int(n)), device=wav.device, dtype=torch.float32)
        L.call("bsed_mel_linear", self._plan, L.ptr(wav), L.c_int(B), L.c_int(n), L.ptr(mel), L.ptr(cmax),
               L.ptr(sumsq), L.ptr(scratch), L.stream())
        return mel, cmax, sumsq

    def stats(self, mel_lin):
        """(clip_max (B,), bin_sumsq (B,n_mels)) of a linear-mel batch (for features loaded from .npy files)"""
        B, T, M = mel_lin.shape
        cmax = torch.empty((B,), device=mel_lin.device, dtype=torch.float32)
        sumsq = torch.empty((B, M), device=mel_lin.device, dtype=torch.float32)
        L.call("bsed_mel_stats", L.ptr(mel_lin), L.c_int(B), L.c_int(T), L.c_int(M), L.ptr(cmax), L.ptr(sumsq),
               L.stream())
        return cmax, sumsq

    def to_db(self, mel_lin, clip_max, max_frames=None):
        B, T, M = mel_lin.shape
        T_out = T if max_frames is None else max_frames
        out = torch.empty((B, 1, T_out, M), device=mel_lin.device, dtype=torch.float32)
        ops._note("mel_db_kernel", f"T{T_out}", 4.0 * B * T_out * M, 4.0 * B * M * (T + T_out))
        L.call("bsed_mel_db", L.ptr(mel_lin), L.ptr(clip_max), L.c_int(B), L.c_int(T), L.c_int(T_out),
               L.c_int(M), L.c_float(self.cfg.top_db), L.ptr(out), L.stream())
        return out

    def add_noise(self, mel_lin, bin_sumsq, seed=0, unit_noise=None):
        B, T, M = mel_lin.shape
        noisy = torch.empty_like(mel_lin)
        cmax = torch.empty((B,), device=mel_lin.device, dtype=torch.float32)
        ops._note("mel_noise_kernel", f"T{T}", 30.0 * B * T * M, 8.0 * B * T * M)
        L.call("bsed_mel_noise", L.ptr(mel_lin), L.ptr(bin_sumsq), L.ptr(unit_noise), L.c_int(B), L.c_int(T),
               L.c_int(M), L.c_float(self.cfg.noise_snr), L.c_u64(seed), L.ptr(noisy), L.ptr(cmax), L.stream())
        return noisy, cmax

    def transform(self, wav, max_frames=None, noisy=False, seed=0, unit_noise=None):
        """waveforms -> dB-mel CRNN input (B,1,max_frames,n_mels) [, noisy twin for the EMA teacher]."""
        max_frames = self.cfg.max_frames if max_frames is None else max_frames
        mel, cmax, sumsq = self.linear(wav)
        clean = self.to_db(mel, cmax, max_frames)
        if not noisy:
            return clean
        nz, nmax = self.add_noise(mel, sumsq, seed=seed, unit_noise=unit_noise)
        return clean, self.to_db(nz, nmax, max_frames)


_default = {}


def preprocess(audio, compute_log=False, cfg=None):
    """Drop-in for the reference ``preprocess``: one waveform (numpy or tensor) -> (T, n_mels) float32
    numpy array of LINEAR mel amplitude (dB if compute_log)."""
    import numpy as np
    cfg = cfg or MelConfig()
    key = (cfg.sr, cfg.n_window, cfg.hop_size, cfg.n_mels, cfg.mel_f_min, cfg.mel_f_max)
    fe = _default.get(key)
    if fe is None:
        fe = _default[key] = MelFrontEnd(cfg)
    wav = torch.as_tensor(np.asarray(audio, dtype=np.float32)).cuda()[None]
    mel, cmax, _ = fe.linear(wav)
    if compute_log:
        mel = fe.to_db(mel, cmax)[:, 0]
    return mel[0].cpu().numpy()
