"""GPU parity of the mel front end (csrc/mel.hip) against the numpy oracle.

Tolerances (fp32 FFT on the GPU vs float64 FFT stored as complex64 in the oracle):
  linear mel : |gpu - ref| <= 2e-5 * clip_max + 2e-5 * |ref|
  dB         : <= 2e-3 dB wherever ref is above the -80 dB clamp floor; exact floor elsewhere
"""
import numpy as np
import pytest
import torch

from oracle import mel_oracle as mo

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def fe():
    from bsed_amd.features import MelFrontEnd
    return MelFrontEnd()


@pytest.fixture(scope="module")
def fe_m():
    """the BASELINE measurement configuration (bench.py): 22.05 kHz, fmax clamped to 11 025 Hz"""
    from bsed_amd.features import MelConfig, MelFrontEnd
    return MelFrontEnd(MelConfig(sr=22050))


def _clips(n, seconds, sr=mo.SR):
    return np.stack([mo.synth_clip(i, sr=sr, seconds=seconds)[0] for i in range(n)])


@pytest.mark.parametrize("seconds", [1.0, 10.0])
@pytest.mark.parametrize("sr", [32000, 22050])
def test_linear_mel_matches_oracle(fe, fe_m, seconds, sr):
    wav = _clips(3, seconds, sr)
    fe = fe if sr == 32000 else fe_m
    assert fe.cfg.mel_f_max == min(16000.0, sr / 2)
    mel, cmax, sumsq = fe.linear(torch.from_numpy(wav).cuda())
    mel, cmax, sumsq = mel.cpu().numpy(), cmax.cpu().numpy(), sumsq.cpu().numpy()
    T = 1 + wav.shape[1] // 255
    assert mel.shape == (3, T, 128)
    for b in range(3):
        ref = mo.preprocess(wav[b], sr=sr, fmax=min(16000.0, sr / 2))
        err = np.abs(mel[b] - ref)
        tol = 2e-5 * ref.max() + 2e-5 * np.abs(ref)
        assert (err <= tol).all(), (err.max(), ref.max(), float((err / tol).max()))
        assert abs(cmax[b] - ref.max()) <= 2e-5 * ref.max()
        np.testing.assert_allclose(sumsq[b], (ref.astype(np.float64) ** 2).sum(0), rtol=1e-4)


@pytest.mark.parametrize("sr,n", [(22050, 220500), (32000, 320000), (22050, 4000), (22050, 255 * 10 + 17),
                                  (22050, 255 * 129 + 100)])
def test_both_stft_kernels(monkeypatch, sr, n):
    """stft_mel2_kernel (two frames per wave, the default) and stft_mel_kernel (one frame per wave, BSED_MEL_PAIR=0 and
    the fallback for filterbanks whose tables do not fit beside eight pair tiles): the same decomposition and operation
    order per frame, fused multiply-adds chosen independently by the compiler in the two bodies -- they agree to a few
    ulp of the clip maximum (measured 2.3e-7), so each is held to the oracle bar by the other's test.  Shapes: the bench
    clip, 32 kHz, a clip of 16 frames that are nearly all edge frames (reflect padding), an odd number of frames (the last
    pair is half empty), and 130 frames = one workgroup of eight waves plus two frames of a second one."""
    from bsed_amd.features import MelConfig, MelFrontEnd
    g = torch.Generator(device="cuda").manual_seed(3)
    wav = (torch.rand(3, n, device="cuda", generator=g) - 0.5) * torch.tensor([0.1, 0.5, 1.0], device="cuda")[:, None]
    outs = []
    for pair in ("1", "0"):
        monkeypatch.setenv("BSED_MEL_PAIR", pair)
        fe = MelFrontEnd(MelConfig(sr=sr))
        assert fe.frames_per_wave == (2 if pair == "1" else 1)
        outs.append([t.cpu().numpy() for t in fe.linear(wav)])
    (m2, c2, s2), (m1, c1, s1) = outs
    for b in range(3):
        assert np.abs(m2[b] - m1[b]).max() <= 2e-6 * m1[b].max()
    np.testing.assert_allclose(c2, c1, rtol=2e-6)
    np.testing.assert_allclose(s2, s1, rtol=1e-5)
    # bitwise repeatable (fixed-order partial sums, no float atomics)
    monkeypatch.setenv("BSED_MEL_PAIR", "1")
    again = [t.cpu().numpy() for t in MelFrontEnd(MelConfig(sr=sr)).linear(wav)]
    assert all(np.array_equal(a, b) for a, b in zip(again, outs[0]))


def test_db_clamp_pad_and_noisy_view(fe):
    wav = _clips(2, 2.0)
    T = 1 + wav.shape[1] // 255
    Tout = T + 9
    rng = np.random.default_rng(5)
    unit = rng.standard_normal((2, T, 128)).astype(np.float32)
    clean, noisy = fe.transform(torch.from_numpy(wav).cuda(), max_frames=Tout, noisy=True,
                                unit_noise=torch.from_numpy(unit).cuda())
    clean, noisy = clean.cpu().numpy(), noisy.cpu().numpy()
    assert clean.shape == (2, 1, Tout, 128)
    for b in range(2):
        ref_c, ref_n = mo.transform_pair(mo.preprocess(wav[b]), Tout, unit_noise=unit[b])
        for got, ref in ((clean[b], ref_c), (noisy[b], ref_n)):
            assert (got[0, T:] == 0).all()
            floor = ref[0, :T].max() - 80.0
            live = ref[0, :T] > floor + 1e-3
            assert np.abs(got[0, :T][live] - ref[0, :T][live]).max() < 2e-3
            if (~live).any():
                assert np.abs(got[0, :T][~live] - ref[0, :T][~live]).max() < 2e-3
            assert got[0, :T].min() >= got[0, :T].max() - 80.0 - 1e-4


def test_bench_config_db_mel_matches_oracle(fe_m):
    """the exact front end bench.py times: 10 s clips at 22.05 kHz -> 865 frames, dB with the per-clip top_db clamp,
    clean and noisy (injected unit noise) views against the oracle's transform_pair"""
    sr = 22050
    wav = _clips(2, 10.0, sr)
    T = 1 + wav.shape[1] // 255
    assert T == 865
    unit = np.random.default_rng(6).standard_normal((2, T, 128)).astype(np.float32)
    clean, noisy = fe_m.transform(torch.from_numpy(wav).cuda(), max_frames=T, noisy=True,
                                  unit_noise=torch.from_numpy(unit).cuda())
    clean, noisy = clean.cpu().numpy(), noisy.cpu().numpy()
    assert clean.shape == (2, 1, T, 128)
    for b in range(2):
        lin = mo.preprocess(wav[b], sr=sr, fmax=sr / 2)
        ref_c, ref_n = mo.transform_pair(lin, T, unit_noise=unit[b])
        for got, ref in ((clean[b], ref_c), (noisy[b], ref_n)):
            assert np.abs(got - ref).max() < 2e-3
            assert got.min() >= got.max() - 80.0 - 1e-4


def test_truncation_and_silence(fe):
    wav = np.zeros((1, 32000), dtype=np.float32)
    out = fe.transform(torch.from_numpy(wav).cuda(), max_frames=100)
    out = out.cpu().numpy()
    assert out.shape == (1, 1, 100, 128)
    assert np.abs(out + 100.0).max() < 1e-3  # all-zero input -> -100 dB everywhere


def test_philox_noise_statistics(fe):
    wav = _clips(1, 10.0)
    mel, cmax, sumsq = fe.linear(torch.from_numpy(wav).cuda())
    nz, _ = fe.add_noise(mel, sumsq, seed=1234)
    d = (nz - mel)[0].double()
    std_ref = torch.sqrt(sumsq[0].double() / mel.shape[1] * 1e-3)
    z = d / std_ref
    assert abs(float(z.mean())) < 0.02 and abs(float(z.std()) - 1.0) < 0.02
    nz2, _ = fe.add_noise(mel, sumsq, seed=1234)
    assert torch.equal(nz, nz2)  # stateless counter RNG: same seed, same draws


def test_preprocess_dropin_signature():
    from bsed_amd.features import preprocess
    y, _ = mo.synth_clip(3, seconds=1.0)
    out = preprocess(y)
    assert out.dtype == np.float32 and out.shape == (1 + len(y) // 255, 128)
