"""Size-independent properties of the HIP path at the BASELINE measurement shape (10 s clips at 22.05 kHz: 865 frames
x 128 mel bands, the reference's batch of 24), where the CPU oracle no longer finishes in seconds.

  * eval-mode forward is per-clip: a batch equals its clips run one by one (BatchNorm uses running statistics);
  * the backward pass is linear in the upstream gradient: G(a*d1 + b*d2) = a*G(d1) + b*G(d2) (same dropout seed);
  * a train step does not depend on the order of the clips in the batch (batch statistics and sums are symmetric);
  * waveform -> mel -> CRNN is the composition of its stages (from_wave=True equals feeding the mel features).
"""
import numpy as np
import pytest
import torch

from oracle import crnn_oracle as co
from oracle import seeded

pytestmark = pytest.mark.gpu

B, T = 24, 865


def _models(dropout, seed=3):
    from bsed_amd.models import CRNN, Predictor, weights_init
    kw = dict(co.CRNN_KWARGS)
    kw["dropout"] = dropout
    torch.manual_seed(seed)
    crnn, pred = CRNN(**kw), Predictor(**co.PREDICTOR_KWARGS)
    weights_init(crnn); weights_init(pred)
    return crnn, pred


def test_eval_forward_is_per_clip_at_full_size():
    crnn, pred = _models(0.5)
    crnn.eval(); pred.eval()
    x = torch.from_numpy(seeded.db_like_input(5, B, T)).cuda()
    with torch.no_grad():
        enc, _ = crnn(x)
        strong, weak = pred(enc)
        assert enc.shape == (B, T // 4, 256) and strong.shape == (B, T // 4, 20) and weak.shape == (B, 20)
        for b in (0, 7, B - 1):
            e1, _ = crnn(x[b:b + 1])
            s1, w1 = pred(e1)
            assert float((e1[0] - enc[b]).abs().max()) < 2e-5
            assert float((s1[0] - strong[b]).abs().max()) < 1e-5 and float((w1[0] - weak[b]).abs().max()) < 1e-5
    assert torch.isfinite(enc).all()


def test_backward_is_linear_in_the_upstream_gradient_at_full_size():
    crnn, _ = _models(0.5)
    crnn.train(); crnn.set_seed(11)
    x = torch.from_numpy(seeded.db_like_input(6, B, T)).cuda()
    g = torch.Generator(device="cuda").manual_seed(1)
    grads = []
    d1 = d2 = None
    for which in range(3):
        enc, ctx = crnn.run_forward(x, save=True)  # same seed -> same dropout masks, same batch statistics
        if d1 is None:
            d1 = torch.randn(enc.shape, device="cuda", generator=g) * 1e-2
            d2 = torch.randn(enc.shape, device="cuda", generator=g) * 1e-2
        d = (d1, d2, 0.5 * d1 - 2.0 * d2)[which]
        crnn.zero_grad()
        crnn.run_backward(ctx, d)
        grads.append(crnn.flat_grad.clone())
    lin = 0.5 * grads[0] - 2.0 * grads[1]
    err = float((grads[2] - lin).norm() / lin.norm())
    assert err < 2e-4, err


def test_train_step_is_invariant_to_clip_order_at_full_size():
    from bsed_amd.engine import FlatAdam, SEDTrainer
    x = torch.from_numpy(seeded.db_like_input(7, B, T)).cuda()
    y = torch.from_numpy(seeded.strong_targets(8, B, T // 4)).cuda()
    perm = torch.randperm(B, generator=torch.Generator().manual_seed(2)).cuda()
    res = []
    for order in (None, perm):
        crnn, pred = _models(0.0)
        tr = SEDTrainer(crnn, pred, optimizer=FlatAdam([crnn, pred], lr=1e-3))
        xx, yy = (x, y) if order is None else (x[order].contiguous(), y[order].contiguous())
        out = tr.train_step(xx, yy)
        res.append((SEDTrainer.loss_value(out), crnn.flat_grad.clone(), pred.flat_grad.clone()))
    assert abs(res[0][0] - res[1][0]) < 1e-5 * abs(res[0][0])
    for i in (1, 2):
        err = float((res[0][i] - res[1][i]).norm() / res[0][i].norm())
        assert err < 1e-4, (i, err)


def test_from_waveform_step_is_the_composition_of_mel_and_crnn_at_full_size():
    from bsed_amd.engine import FlatAdam, SEDTrainer
    from bsed_amd.features import MelConfig, MelFrontEnd
    sr, nb = 22050, 8
    rng = np.random.default_rng(4)
    wav = torch.from_numpy((0.1 * rng.standard_normal((nb, 10 * sr))).astype(np.float32)).cuda()
    fe = MelFrontEnd(MelConfig(sr=sr))
    frames = fe.num_frames(wav.shape[1])
    assert frames == T
    y = torch.from_numpy(seeded.strong_targets(9, nb, T // 4)).cuda()
    losses = []
    for from_wave in (True, False):
        crnn, pred = _models(0.0)
        tr = SEDTrainer(crnn, pred, optimizer=FlatAdam([crnn, pred], lr=1e-3), frontend=fe)
        inp = wav if from_wave else fe.transform(wav, max_frames=T)
        out = tr.train_step(inp, y, from_wave=from_wave)
        losses.append((SEDTrainer.loss_value(out), crnn.flat_grad.clone()))
    assert abs(losses[0][0] - losses[1][0]) < 1e-6 * abs(losses[0][0])
    assert torch.equal(losses[0][1], losses[1][1])


def test_train_step_is_bitwise_repeatable_at_full_size():
    """Two runs from the same state give bit-identical gradients (dropout on): every reduction in the path is ordered
    (partial slabs + fixed-order sums, no float atomics on the CRNN path).  This check found a VALU -> MFMA SrcC
    hand-off hazard (one stale element per ~10 runs) that no tolerance-based test could see."""
    from bsed_amd.engine import FlatAdam, SEDTrainer
    x = torch.from_numpy(seeded.db_like_input(12, B, T)).cuda()
    y = torch.from_numpy(seeded.strong_targets(13, B, T // 4)).cuda()
    grads = []
    for rep in range(3):
        crnn, pred = _models(0.5)
        crnn.set_seed(5)
        tr = SEDTrainer(crnn, pred, optimizer=FlatAdam([crnn, pred], lr=1e-3), seed=7)
        junk = torch.empty((rep + 1) << 20, device="cuda")  # shift the allocator between repetitions
        tr.train_step(x, y)
        grads.append((crnn.flat_grad.clone(), pred.flat_grad.clone()))
        del junk
    for rep in (1, 2):
        assert torch.equal(grads[0][0], grads[rep][0]) and torch.equal(grads[0][1], grads[rep][1])


@pytest.mark.parametrize("Bq,Tq,Fq,seed", [(1, 64, 128, 1), (3, 100, 128, 2), (5, 260, 128, 3), (2, 1255, 128, 4), (7, 36, 128, 5)])
def test_split_fp32_and_fp32_modes_agree_on_odd_shapes(Bq, Tq, Fq, seed):
    """the two contraction modes are independent kernel families (bf16 cores with split operands vs fp32 cores):
    their forward outputs and every gradient tensor must agree on shapes with partial tiles, odd pooled extents and
    batch sizes that leave idle rows in the recurrence workgroups.  Dropout on (same masks by construction)."""
    outs = []
    x = torch.from_numpy(seeded.db_like_input(20 + seed, Bq, Tq, Fq)).cuda()
    for mode in ("bf16x3", "fp32"):
        crnn, _ = _models(0.5, seed=seed)
        crnn.conv_mode = mode
        crnn.train(); crnn.set_seed(3)
        enc, ctx = crnn.run_forward(x, save=True)
        d = (torch.cos(torch.arange(enc.numel(), device="cuda", dtype=torch.float32)).view_as(enc) * 1e-2)
        crnn.zero_grad(); crnn._attach_grads()
        crnn.run_backward(ctx, d)
        outs.append((enc.clone(), {k: p.grad.clone() for k, p in crnn.named_parameters()}))
    assert float((outs[0][0] - outs[1][0]).abs().max()) < 1e-4
    bad = []
    for k in outs[0][1]:
        a, b = outs[0][1][k], outs[1][1][k]
        err = float((a - b).norm() / (b.norm() + 1e-12))
        if err > 3e-4 and float(b.norm()) > 1e-7:
            bad.append((k, err))
    assert not bad, bad
