"""The CPU oracle must reproduce the vectors produced by the reference itself
(tests/golden/*, written by oracle/gen_golden.py in the build container)."""
import json
import os

import numpy as np
import pytest
import torch

from oracle import crnn_oracle as co
from oracle import labels_oracle as lo
from oracle import mel_oracle as mo
from oracle import seeded


def _load(golden_dir, name):
    return np.load(os.path.join(golden_dir, name), allow_pickle=False)


def _build(dropout, seed):
    kw = dict(co.CRNN_KWARGS)
    kw["dropout"] = dropout
    crnn, pred = co.CRNN(**kw), co.Predictor(**co.PREDICTOR_KWARGS)
    v1 = seeded.load_seeded(crnn, seed)
    v2 = seeded.load_seeded(pred, seed + 1)
    return crnn, pred, seeded.checksum(v1), seeded.checksum(v2)


@pytest.mark.parametrize("tag", ["small", "R"])
def test_crnn_forward_matches_reference(golden_dir, tag):
    g = _load(golden_dir, f"crnn_{tag}.npz")
    B, T, seed = (int(v) for v in g["meta"])
    x = torch.from_numpy(seeded.db_like_input(seed + 10, B, T))
    crnn, pred, c1, c2 = _build(0.5, seed)
    assert np.allclose([c1, c2], g["weight_checksum"], rtol=0, atol=0)
    crnn.eval(); pred.eval()
    with torch.no_grad():
        enc, _ = crnn(x)
        strong, weak = pred(enc)
    # same torch build, same ops -> bitwise on one machine; 1e-6 across CPUs
    np.testing.assert_allclose(enc.numpy(), g["eval_enc"], atol=2e-6)
    np.testing.assert_allclose(strong.numpy(), g["eval_strong"], atol=1e-6)
    np.testing.assert_allclose(weak.numpy(), g["eval_weak"], atol=1e-6)


def test_crnn_train_step_matches_reference(golden_dir):
    g = _load(golden_dir, "crnn_small.npz")
    B, T, seed = (int(v) for v in g["meta"])
    x = torch.from_numpy(seeded.db_like_input(seed + 10, B, T))
    y = torch.from_numpy(seeded.strong_targets(seed + 11, B, T // 4))
    crnn, pred, _, _ = _build(0.0, seed)
    crnn.train(); pred.train()
    opt = torch.optim.Adam(list(crnn.parameters()) + list(pred.parameters()), lr=1e-3)
    losses = []
    for step in range(len(g["train_losses"])):
        opt.zero_grad()
        loss, out = co.train_losses(crnn, pred, x, y)
        loss.backward()
        if step == 0:
            np.testing.assert_allclose(out["strong_syn"].detach().numpy(), g["train_strong"], atol=1e-6)
            names = list(g["grad_names"])
            grads = {"crnn." + k: p.grad for k, p in crnn.named_parameters()}
            grads.update({"pred." + k: p.grad for k, p in pred.named_parameters()})
            # the oracle's parameter names carry the un-stripped "cnn.cnn." level
            norms = np.array([float(grads[n].double().norm()) for n in names])
            np.testing.assert_allclose(norms, g["grad_norms"], rtol=1e-4, atol=1e-9)
        opt.step()
        losses.append(float(loss))
    np.testing.assert_allclose(losses, g["train_losses"], rtol=1e-5)


def test_mean_teacher_loss_and_ema(golden_dir):
    g = _load(golden_dir, "crnn_small.npz")
    B, T, seed = (int(v) for v in g["meta"])
    x = torch.from_numpy(seeded.db_like_input(seed + 10, B, T))
    y = torch.from_numpy(seeded.strong_targets(seed + 11, B, T // 4))
    crnn, pred, _, _ = _build(0.0, seed)
    ema_c, ema_p, _, _ = _build(0.0, seed + 5)
    for m in (crnn, pred, ema_c, ema_p):
        m.train()
    xr = seeded.db_like_input(seed + 20, B, T)
    xe = xr + np.random.default_rng(seed + 21).normal(0, 1.0, xr.shape).astype(np.float32)
    yw = (np.random.default_rng(seed + 22).random((B, 20)) < 0.2).astype(np.float32)
    opt = torch.optim.Adam(list(crnn.parameters()) + list(pred.parameters()), lr=1e-3)
    opt.zero_grad()
    loss, out = co.train_losses(crnn, pred, x, y, torch.from_numpy(xr), torch.from_numpy(yw),
                                ema=(ema_c, ema_p), x_real_ema=torch.from_numpy(xe), consistency_cost=0.7)
    loss.backward()
    opt.step()
    assert abs(float(loss) - float(g["mt_loss"])) < 1e-5
    np.testing.assert_allclose(out["strong_ema"].numpy(), g["mt_strong_ema"], atol=1e-6)
    for gs in (1, 5000):
        co.update_ema_variables(crnn, ema_c, 0.999, gs)
        co.update_ema_variables(pred, ema_p, 0.999, gs)
        sd = {"crnn." + k: v for k, v in ema_c.state_dict().items()}
        sd.update({"pred." + k: v for k, v in ema_p.state_dict().items()})
        names = list(g[f"ema{gs}_names"])
        # A conv bias feeding train-mode BatchNorm has an exactly-zero gradient in exact
        # arithmetic; Adam turns the round-off there into +-lr steps, so those entries are noise.
        keep = np.array([not (".conv" in n and n.endswith(".bias")) for n in names])
        norms = np.array([float(sd[n].double().norm()) for n in names])
        np.testing.assert_allclose(norms[keep], g[f"ema{gs}_norms"][keep], rtol=1e-5, atol=1e-7)
        for n in np.array(names)[keep]:
            key = f"ema{gs}/{n}"
            if key in g.files:
                np.testing.assert_allclose(sd[n].numpy(), g[key], rtol=1e-5, atol=1e-5)  # Adam normalises round-off-sized grads


def test_isp_loss_assembly_matches_reference_golden(golden_dir):
    """oracle.train_losses_isp against tests/golden/isp.npz, which oracle/gen_golden.py::isp_case assembled from the
    IMPORTED reference CRNN / Predictor with the reference's own per-sample torch.roll loops and loss composition
    (src/main_baseline.py:229-277,337-420,442-529): loss, and every parameter gradient."""
    g = _load(golden_dir, "isp.npz")
    B, T, seed = (int(v) for v in g["meta"])
    rng = np.random.default_rng(seed)
    xs = seeded.db_like_input(seed + 1, B, T); xr = seeded.db_like_input(seed + 2, B, T)
    xe = xr + rng.normal(0, 1.0, xr.shape).astype(np.float32)
    y = seeded.strong_targets(seed + 3, B, T // 4)
    yw = (rng.random((B, 20)) < 0.2).astype(np.float32)
    crnn, pred, c1, c2 = _build(0.0, seed)
    ema_c, ema_p, c3, c4 = _build(0.0, seed + 5)
    assert np.array_equal([c1, c2, c3, c4], g["weight_checksum"])
    for m in (crnn, pred, ema_c, ema_p):
        m.train()
    tt = torch.from_numpy
    loss = co.train_losses_isp(crnn, pred, (ema_c, ema_p), tt(xs), tt(y), tt(xr), tt(yw), tt(xe),
                               [int(v) for v in g["shift_frames"]], [int(v) for v in g["shift_bins"]],
                               consistency_cost=float(g["consistency_cost"]))
    loss.backward()
    assert abs(float(loss.detach()) - float(g["loss"])) < 1e-5 * abs(float(g["loss"]))
    grads = {"crnn." + k: p.grad for k, p in crnn.named_parameters()}
    grads.update({"pred." + k: p.grad for k, p in pred.named_parameters()})
    names = list(g["grad_names"])
    norms = np.array([float(grads[n].double().norm()) for n in names])
    # a conv bias feeding train-mode BatchNorm has an exactly-zero gradient: those rows are round-off on both sides
    keep = np.array([not (".conv" in n and n.endswith(".bias")) for n in names])
    np.testing.assert_allclose(norms[keep], g["grad_norms"][keep], rtol=1e-4, atol=1e-9)
    for n in np.array(names)[keep]:
        if "grad/" + n in g.files:
            # per tensor, relative L2: the two assemblies add the nine passes' gradients in different orders (fp32)
            ref = g["grad/" + n].astype(np.float64)
            assert np.linalg.norm(grads[n].numpy() - ref) <= 1e-4 * np.linalg.norm(ref) + 1e-9, n


def test_clip_discriminator_and_domain_loss(golden_dir):
    g = _load(golden_dir, "clipd.npz")
    B, T, seed = (int(v) for v in g["meta"])
    rng = np.random.default_rng(seed)
    f_s = rng.standard_normal((B, T, 256)).astype(np.float32)
    f_t = rng.standard_normal((B, T, 256)).astype(np.float32)
    disc = co.Clip_Discriminator()
    vals = seeded.load_seeded(disc, seed + 1)
    assert seeded.checksum(vals) == float(g["weight_checksum"])
    disc.train()
    for it in range(3):
        fs = torch.from_numpy(f_s).requires_grad_()
        ft = torch.from_numpy(f_t).requires_grad_()
        disc.zero_grad()
        loss = co.domain_loss(disc, fs, ft, co.grl_coeff(it))
        loss.backward()
        assert abs(float(loss) - float(g[f"loss{it}"])) < 1e-6
        assert abs(float(fs.grad.norm()) - float(g[f"dfs_norm{it}"])) < 1e-6 + 1e-4 * float(g[f"dfs_norm{it}"])
        np.testing.assert_allclose(fs.grad.numpy()[:, ::16, ::8], g[f"dfs{it}"], atol=1e-7, rtol=1e-4)
    disc.eval()
    with torch.no_grad():
        out = disc(torch.from_numpy(np.concatenate([f_s, f_t])))
    np.testing.assert_allclose(out.numpy(), g["eval_out"], atol=1e-6)


def test_weights_init_statistics(golden_dir):
    ref = json.load(open(os.path.join(golden_dir, "weights_init.json")))
    crnn, pred = co.CRNN(**co.CRNN_KWARGS), co.Predictor(**co.PREDICTOR_KWARGS)
    torch.manual_seed(2023)
    crnn.apply(co.weights_init); pred.apply(co.weights_init)
    stats = {}
    for pfx, m in (("crnn.", crnn), ("pred.", pred)):
        for k, v in m.state_dict().items():
            stats[pfx + k] = v.double()
    for k, val in ref.items():
        if k.startswith("_"):
            continue
        mean, std, amax, asum = val
        v = stats[k]
        if ".bias_ih_" in k or ".bias_hh_" in k:
            # weights_init leaves GRU biases alone: they keep the constructor's U(-1/sqrt(H), 1/sqrt(H)) draw
            assert float(v.abs().max()) <= 1 / np.sqrt(128) and amax <= 1 / np.sqrt(128)
            continue
        # same seed + same traversal order => same draws (LAPACK-dependent last bits for orthogonal_)
        assert abs(float(v.mean()) - mean) < 1e-5 + 1e-4 * abs(mean), k
        assert abs(float(v.abs().sum()) - asum) < 1e-3 + 1e-4 * asum, k
    w = stats["crnn.rnn.rnn.weight_hh_l0"]
    assert float((w.T @ w - torch.eye(128, dtype=torch.double)).abs().max()) < 1e-5


def test_schedules(golden_dir):
    ref = json.load(open(os.path.join(golden_dir, "schedules.json")))
    assert np.allclose([co.sigmoid_rampdown(e, 30) for e in range(0, 40, 3)], ref["sigmoid_rampdown_30"], rtol=0, atol=0)
    assert np.allclose([co.exp_rampup(e, 50) for e in range(0, 60, 5)], ref["exp_rampup_50"], rtol=0, atol=0)
    for e, lr, lr_d, lr_c in ref["adjust_learning_rate"]:
        mine = co.learning_rate(co.sigmoid_rampdown(e, 30), e, 0.0005)
        assert mine == lr and mine * 0.1 == lr_d and mine * 0.1 == lr_c
    assert np.allclose([co.grl_coeff(i) for i in range(5)], ref["grl_coeff_first5"], atol=1e-7)


def test_label_frame_indexing(golden_dir):
    cases = json.load(open(os.path.join(golden_dir, "labels_kat.json")))
    for c in cases:
        y = lo.encode_strong([(on, off, lab) for on, off, lab in c["events"]], 313)
        assert float(y.sum()) == c["sum"]
        assert y.sum(0).tolist() == c["col_sums"]
        assert lo.encode_weak([e[2] for e in c["events"]]).tolist() == c["weak"]
    # the survey's known answers (clip 00.wav)
    assert (lo.frame_index(3.279), lo.frame_index(4.463)) == (102, 140)
    assert (lo.frame_index(4.888), lo.frame_index(6.272)) == (153, 196)
    assert (lo.frame_index(6.549), lo.frame_index(7.476)) == (205, 234)
    assert (lo.frame_index(6.550), lo.frame_index(8.213)) == (205, 257)


def test_decode_and_post_process_roundtrip():
    y = lo.encode_strong([(1.0, 3.0, "EATO"), (5.0, 5.15, "AMCR"), (2.0, 9.99, "BAWW")], 313)
    ev = lo.decode_strong(y)
    assert sorted(ev) == sorted([["EATO", lo.frame_index(1.0), lo.frame_index(3.0)],
                                 ["AMCR", lo.frame_index(5.0), lo.frame_index(5.15)],
                                 ["BAWW", lo.frame_index(2.0), lo.frame_index(9.99)]])
    post = lo.post_process(y * 0.9, median_window=14)
    labs = [p[0] for p in post]
    assert "EATO" in labs and "BAWW" in labs and "AMCR" not in labs  # 4-frame event loses the 14-frame median vote
    for _, on, off in post:
        assert 0.0 <= on < off <= 10.0


# ---------------------------------------------------------------- mel stage: self-pins (parity unpinned)
def test_mel_known_answers():
    kat = [0, 85.317, 170.635, 255.952, 341.269, 426.586, 511.904, 597.221, 682.538, 767.855,
           853.173, 938.49, 1024.856, 1119.114]
    np.testing.assert_allclose(mo.mel_frequencies(40, 0, 11025)[:14], kat, atol=2e-3)
    fb = mo.mel_filterbank()
    assert fb.shape == (128, 1025) and fb.dtype == np.float32
    assert int((fb > 0).sum()) == 2016
    assert abs(float(fb[0].sum()) - 1.814153) < 1e-5
    assert abs(np.hamming(2048).sum() - 1105.46) < 1e-2


def test_mel_sinusoid_silence_and_clamp():
    sr, n = 32000, 32000
    k = 200  # bin-centred
    t = np.arange(n) / sr
    y = np.sin(2 * np.pi * (k * sr / 2048) * t).astype(np.float32)
    S = mo.stft_mag(y)
    assert S.shape == (1025, 1 + n // 255)
    assert abs(S[k, 60] - 552.73) < 0.05
    mel = mo.preprocess(np.zeros(n, dtype=np.float32))
    assert mel.shape == (126, 128) and mel.dtype == np.float32
    db = mo.amplitude_to_db(mel.T).T
    assert np.all(np.abs(db + 100.0) < 1e-4)  # float32(1e-10) is not exactly 1e-10
    y2, _ = mo.synth_clip(0, seconds=1.0)
    db2 = mo.amplitude_to_db(mo.preprocess(y2).T).T
    assert db2.min() >= db2.max() - 80.0 - 1e-4
    clean, noisy = mo.transform_pair(mo.preprocess(y2), 130, unit_noise=np.zeros((126, 128)))
    assert clean.shape == (1, 130, 128) and np.all(clean[0, 126:] == 0)
    np.testing.assert_allclose(clean, noisy, atol=1e-4)


def test_oracle_crnn_fpn_matches_reference_golden(golden_dir):
    """oracle.crnn_oracle.CRNN_fpn (restatement of src/models/CRNN_GRL.py:293-389 + CNN_FPN.py:33-100) against the
    vectors the reference's own CRNN_fpn produced: same state-dict entries, eval and train forward, gradient norms"""
    import os
    import numpy as np
    import torch
    from oracle import crnn_oracle as co, seeded
    g = np.load(os.path.join(golden_dir, "crnn_fpn.npz"))
    B, T, seed = (int(v) for v in g["meta"])
    kw = dict(co.CRNN_KWARGS)
    kw["dropout"] = 0.5
    m = co.CRNN_fpn(**kw)
    assert list(m.state_dict().keys()) == [str(n) for n in g["state_names"]]
    vals = seeded.load_seeded(m, seed)
    assert abs(seeded.checksum(vals) - float(g["weight_checksum"][0])) < 1e-6 * float(g["weight_checksum"][0])
    x = torch.from_numpy(seeded.db_like_input(seed + 10, B, T))
    m.eval()
    with torch.no_grad():
        enc, _ = m(x)
    np.testing.assert_allclose(enc.numpy(), g["eval_enc"], atol=1e-6)
    kw["dropout"] = 0.0
    m = co.CRNN_fpn(**kw)
    seeded.load_seeded(m, seed)
    m.cnn.dropout.p = 0.0
    m.train()
    enc, _ = m(x)
    np.testing.assert_allclose(enc.detach().numpy(), g["train_enc"], atol=1e-5)
    w = torch.sin(torch.arange(enc.numel(), dtype=torch.float32)).view_as(enc) * 1e-2
    (enc * w).sum().backward()
    grads = {k: p.grad for k, p in m.named_parameters() if p.grad is not None}
    for n, ref_norm in zip(g["grad_names"], g["grad_norms"]):
        assert abs(float(grads[str(n)].double().norm()) - ref_norm) < 1e-4 * ref_norm + 1e-9, str(n)


# ---------------------------------------------------------------- CRNN_pred (BASELINE configs[1]) and transforms
def _cnn_pred_kwargs(dropout):
    kw = dict(co.CRNN_KWARGS)
    kw.update(nclass=128, n_RNN_cell=64, dropout=dropout)
    return kw


@pytest.mark.parametrize("tag", ["small", "R"])
def test_oracle_cnn_pred_matches_reference_golden(golden_dir, tag):
    g = _load(golden_dir, "cnn_pred.npz")
    B, T, seed = (int(v) for v in g[f"{tag}_meta"])
    x = torch.from_numpy(seeded.db_like_input(seed + 10, B, T))
    m = co.CRNN_pred(**_cnn_pred_kwargs(0.5))
    assert list(m.state_dict().keys()) == [str(n) for n in g["state_names"]]
    vals = seeded.load_seeded(m, seed)
    assert seeded.checksum(vals) == float(g[f"{tag}_weight_checksum"][0])
    m.eval()
    with torch.no_grad():
        strong, weak = m(x)
    np.testing.assert_allclose(strong.numpy(), g[f"{tag}_eval_strong"], atol=1e-6)
    np.testing.assert_allclose(weak.numpy(), g[f"{tag}_eval_weak"], atol=1e-6)
    m = co.CRNN_pred(**_cnn_pred_kwargs(0.0))
    seeded.load_seeded(m, seed)
    m.train()
    with torch.no_grad():
        strong, weak = m(x)
    np.testing.assert_allclose(strong.numpy(), g[f"{tag}_train_strong"], atol=2e-6)
    np.testing.assert_allclose(weak.numpy(), g[f"{tag}_train_weak"], atol=2e-6)
    np.testing.assert_allclose(m.state_dict()["cnn.batchnorm6.running_var"].numpy(), g[f"{tag}_after_rv6"], rtol=1e-5)


def test_oracle_transforms_match_reference_golden(golden_dir):
    """a2 / a4: the restated gaussian_noise / pad_trunc_seq / pipeline against vectors produced by RUNNING the
    reference's src/data/Transforms.py (numpy legacy RNG seeded identically)."""
    g = _load(golden_dir, "transforms.npz")
    x = g["x"]
    np.random.seed(2023)
    got = mo.gaussian_noise(x, 30.0, rng=np.random)
    assert got.dtype == np.float64
    np.testing.assert_allclose(got, g["noisy_snr30"], rtol=0, atol=1e-12)
    std = mo.gaussian_noise_std(x, 30.0)
    np.testing.assert_allclose(std, np.sqrt(np.mean((x.astype(np.float32) ** 2) * 1e-3, axis=0)), rtol=1e-6)
    for key, n in (("pad50", 50), ("trunc20", 20), ("same37", 37)):
        out = mo.pad_trunc_seq(x, n)
        assert out.shape == g[key].shape and np.array_equal(out, g[key])
    x3 = np.stack([x, 0.5 * x])
    assert np.array_equal(mo.pad_trunc_seq(x3, 40), g["pad3d_40"])
    for tag, frames in (("pad", 50), ("trunc", 20)):
        np.random.seed(99)
        clean, noisy = mo.transform_pair(x, frames, snr=30.0, unit_noise=np.random.normal(0, 1, x.shape))
        assert clean.shape == g[f"pipe_{tag}_clean"].shape and clean.dtype == np.float32
        np.testing.assert_allclose(clean, g[f"pipe_{tag}_clean"], rtol=0, atol=1e-6)
        np.testing.assert_allclose(noisy, g[f"pipe_{tag}_noisy"], rtol=0, atol=1e-5)


def test_oracle_fpn_weights_init_statistics(golden_dir):
    ref = json.load(open(os.path.join(golden_dir, "weights_init_fpn.json")))
    m = co.CRNN_fpn(**co.CRNN_KWARGS)
    torch.manual_seed(2023)
    m.apply(co.weights_init)
    sd = m.state_dict()
    assert sorted(sd.keys()) == sorted(ref.keys())
    for k, (mean, std, amax, asum) in ref.items():
        v = sd[k].double()
        if ".bias_ih_" in k or ".bias_hh_" in k:
            assert float(v.abs().max()) <= 1 / np.sqrt(128) and amax <= 1 / np.sqrt(128)
            continue
        assert abs(float(v.mean()) - mean) < 1e-5 + 1e-4 * abs(mean), k
        assert abs(float(v.abs().sum()) - asum) < 1e-3 + 1e-4 * asum, k


def test_oracle_frame_discriminator_matches_reference_golden(golden_dir):
    g = _load(golden_dir, "frame_d.npz")
    N, T, seed = (int(v) for v in g["meta"])
    x = torch.from_numpy(np.random.default_rng(seed).standard_normal((N, T, 256)).astype(np.float32)).requires_grad_()
    m = co.Frame_Discriminator(input_dim=256, dropout=0)
    assert list(m.state_dict().keys()) == [str(n) for n in g["state_names"]]
    vals = seeded.load_seeded(m, seed + 1)
    assert seeded.checksum(vals) == float(g["weight_checksum"][0])
    m.train()
    d = m(x)
    np.testing.assert_allclose(d.detach().numpy(), g["out"], atol=1e-6)
    up = torch.cos(torch.arange(d.numel(), dtype=torch.float32)).view_as(d) * 0.3
    (d * up).sum().backward()
    np.testing.assert_allclose(x.grad.numpy()[:, ::7, ::5], g["dx"], atol=1e-7, rtol=1e-4)
    for k, p in m.named_parameters():
        np.testing.assert_allclose(p.grad.numpy(), g["grad/" + k], atol=1e-6, rtol=1e-4)


@pytest.mark.skipif(not os.path.isdir("/root/reference/src"), reason="the reference only exists in the build container")
def test_committed_fixtures_are_what_the_generator_writes():
    """tools/check_golden.py: regenerate every fixture from the reference into a scratch directory and compare bit for bit"""
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    r = subprocess.run([sys.executable, os.path.join(root, "tools", "check_golden.py")], capture_output=True, text=True)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]
