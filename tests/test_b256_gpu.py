"""The BASELINE measurement configuration itself: B = 256 clips of 865 frames x 128 mel bands (10 s at 22.05 kHz) on one
MI355X.  Every other GPU test stops at the reference's batch of 24; at 256 the persistent-grid caps, the head splits,
the GRU row grouping and the weight-gradient slab counts take other branches and the first-block tensors are 1.8 GB.

  * eval-mode outputs of two of the 256 clips against the CPU oracle (tolerances of test_crnn_gpu.py);
  * eval-mode forward is per-clip (batch == the clip run alone);
  * a train step (dropout 0.5) is bitwise repeatable and invariant to the clip order;
  * the two contraction modes (split-fp32 on the bf16 cores, exact fp32 cores) agree on the encoder output and on
    every gradient tensor;
  * the from-waveform step equals the composition mel -> step (bitwise).
"""
import numpy as np
import pytest
import torch

from oracle import crnn_oracle as co
from oracle import seeded

pytestmark = pytest.mark.gpu

B, T = 256, 865


@pytest.fixture(scope="module")
def batch():
    x = torch.from_numpy(seeded.db_like_input(41, B, T)).cuda()
    y = torch.from_numpy(seeded.strong_targets(42, B, T // 4)).cuda()
    return x, y


def _oracle_pair(seed):
    kw = dict(co.CRNN_KWARGS)
    ocrnn, opred = co.CRNN(**kw), co.Predictor(**co.PREDICTOR_KWARGS)
    seeded.load_seeded(ocrnn, seed); seeded.load_seeded(opred, seed + 1)
    return ocrnn, opred


def _product_pair(ocrnn, opred, dropout):
    from bsed_amd.models import CRNN, Predictor
    kw = dict(co.CRNN_KWARGS); kw["dropout"] = dropout
    crnn, pred = CRNN(**kw), Predictor(**co.PREDICTOR_KWARGS)
    crnn.load_state_dict(ocrnn.state_dict()); pred.load_state_dict(opred.state_dict())
    return crnn, pred


def test_eval_forward_at_b256_matches_oracle_and_is_per_clip(batch):
    x, _ = batch
    ocrnn, opred = _oracle_pair(61)
    crnn, pred = _product_pair(ocrnn, opred, 0.5)
    ocrnn.eval(); opred.eval(); crnn.eval(); pred.eval()
    with torch.no_grad():
        enc, _ = crnn(x)
        strong, weak = pred(enc)
        assert enc.shape == (B, T // 4, 256) and bool(torch.isfinite(enc).all())
        clips = [5, 200]
        xe = x[clips].cpu()
        enc_o, _ = ocrnn(xe)
        strong_o, weak_o = opred(enc_o)
        for i, b in enumerate(clips):
            assert float((enc[b].cpu() - enc_o[i]).abs().max()) < 1e-4
            assert float((strong[b].cpu() - strong_o[i]).abs().max()) < 2e-5
            assert float((weak[b].cpu() - weak_o[i]).abs().max()) < 2e-5
        for b in (0, 131, B - 1):
            e1, _ = crnn(x[b:b + 1])
            s1, w1 = pred(e1)
            assert float((e1[0] - enc[b]).abs().max()) < 2e-5
            assert float((s1[0] - strong[b]).abs().max()) < 1e-5 and float((w1[0] - weak[b]).abs().max()) < 1e-5


def test_train_step_at_b256_is_bitwise_repeatable_and_order_invariant(batch):
    from bsed_amd.engine import FlatAdam, SEDTrainer
    x, y = batch
    ocrnn, opred = _oracle_pair(62)
    perm = torch.randperm(B, generator=torch.Generator().manual_seed(3)).cuda()
    res = []
    for rep, (order, drop) in enumerate(((None, 0.5), (None, 0.5), (None, 0.0), (perm, 0.0))):
        crnn, pred = _product_pair(ocrnn, opred, drop)
        tr = SEDTrainer(crnn, pred, optimizer=FlatAdam([crnn, pred], lr=1e-3), seed=7)
        junk = torch.empty((rep + 1) << 22, device="cuda")
        xx, yy = (x, y) if order is None else (x[order].contiguous(), y[order].contiguous())
        out = tr.train_step(xx, yy)
        res.append((SEDTrainer.loss_value(out), crnn.flat_grad.clone(), pred.flat_grad.clone(), crnn.flat.clone()))
        del junk
    assert np.isfinite(res[0][0]) and res[0][0] > 0
    assert res[0][0] == res[1][0]
    for i in (1, 2, 3):
        assert torch.equal(res[0][i], res[1][i]), i
    assert abs(res[2][0] - res[3][0]) < 1e-5 * abs(res[2][0])
    for i in (1, 2):
        err = float((res[2][i] - res[3][i]).norm() / res[2][i].norm())
        assert err < 1e-4, (i, err)


def test_contraction_modes_agree_at_b256(batch):
    x, _ = batch
    ocrnn, opred = _oracle_pair(63)
    outs = []
    for mode in ("bf16x3", "fp32"):
        crnn, _ = _product_pair(ocrnn, opred, 0.5)
        crnn.conv_mode = mode
        crnn.train(); crnn.set_seed(9)
        enc, ctx = crnn.run_forward(x, save=True)
        d = torch.cos(torch.arange(enc.numel(), device="cuda", dtype=torch.float32)).view_as(enc) * 1e-3
        crnn.zero_grad(); crnn._attach_grads()
        crnn.run_backward(ctx, d)
        del ctx
        outs.append((enc.clone(), {k: p.grad.clone() for k, p in crnn.named_parameters()}))
        del crnn
        torch.cuda.empty_cache()
    assert float((outs[0][0] - outs[1][0]).abs().max()) < 1e-4
    bad = []
    for k in outs[0][1]:
        a, b_ = outs[0][1][k], outs[1][1][k]
        err = float((a - b_).norm() / (b_.norm() + 1e-20))
        if err > 3e-4 and float(b_.norm()) > 1e-9:
            bad.append((k, err))
    assert not bad, bad


def test_from_waveform_step_at_b256_is_the_composition(batch):
    from bsed_amd.engine import FlatAdam, SEDTrainer
    from bsed_amd.features import MelConfig, MelFrontEnd
    _, y = batch
    sr = 22050
    g = torch.Generator(device="cuda").manual_seed(4)
    wav = torch.randn(B, 10 * sr, device="cuda", generator=g) * 0.1
    t = torch.arange(10 * sr, device="cuda", dtype=torch.float32) / sr
    wav += 0.3 * torch.sin(2 * np.pi * (500.0 + 37.0 * torch.arange(B, device="cuda")[:, None]) * t[None])
    fe = MelFrontEnd(MelConfig(sr=sr))
    assert fe.num_frames(wav.shape[1]) == T
    ocrnn, opred = _oracle_pair(64)
    res = []
    for from_wave in (True, False):
        crnn, pred = _product_pair(ocrnn, opred, 0.5)
        tr = SEDTrainer(crnn, pred, optimizer=FlatAdam([crnn, pred], lr=1e-3), frontend=fe, seed=5)
        inp = wav if from_wave else fe.transform(wav, max_frames=T)
        out = tr.train_step(inp, y, from_wave=from_wave)
        res.append((SEDTrainer.loss_value(out), crnn.flat_grad.clone()))
    assert np.isfinite(res[0][0]) and res[0][0] == res[1][0]
    assert torch.equal(res[0][1], res[1][1])


def test_mean_teacher_and_adversarial_steps_at_bench_shape_are_repeatable(batch):
    """BASELINE configs[3] / [4] as bench.py runs them (--mode mt / ada): 128 synthetic + 128 real clips of 865 frames,
    dropout 0.5.  Two runs from the same state: identical losses and gradient arenas, bit for bit (student, predictor
    and discriminator gradients share one arena); the EMA teacher moves by (1 - alpha_t) of the gap; the adversarial
    step's domain loss starts at the discriminator's chance level and its gradient reaches the CRNN."""
    from bsed_amd.disc import Clip_Discriminator, ConditionalDomainAdversarialLoss
    from bsed_amd.engine import FlatAdam, FlatSGD, SEDTrainer
    x, y = batch
    h = B // 2
    xs, ys, xr = x[:h].contiguous(), y[:h].contiguous(), x[h:].contiguous()
    yw = y[h:].max(1)[0].contiguous()
    xe = (xr + 0.5).contiguous()
    ocrnn, opred = _oracle_pair(65)
    # ---- mean teacher
    runs = []
    for rep in range(2):
        crnn, pred = _product_pair(ocrnn, opred, 0.5)
        ema_c, ema_p = _product_pair(ocrnn, opred, 0.5)
        tr = SEDTrainer(crnn, pred, ema_c, ema_p, optimizer=FlatAdam([crnn, pred], lr=1e-3), seed=9)
        before = crnn.flat.clone()
        out = tr.train_step(xs, ys, xr, yw, xe)
        runs.append((SEDTrainer.loss_value(out), tr.arena.flat.clone(), crnn.flat.clone(), ema_c.flat.clone()))
        want = 0.5 * before + 0.5 * crnn.flat        # alpha = min(1 - 1/(1+1), 0.999) = 0.5 at global step 1
        assert float((ema_c.flat - want).abs().max()) <= 1e-6 * float(want.abs().max()) + 1e-9
    assert np.isfinite(runs[0][0]) and runs[0][0] == runs[1][0]
    for i in (1, 2, 3):
        assert torch.equal(runs[0][i], runs[1][i]), i
    # ---- domain adversarial
    runs = []
    for rep in range(2):
        crnn, pred = _product_pair(ocrnn, opred, 0.5)
        torch.manual_seed(3)
        disc = Clip_Discriminator()
        cdan = ConditionalDomainAdversarialLoss(disc)
        cdan.iter_num = 500
        tr = SEDTrainer(crnn, pred, optimizer=FlatSGD([crnn, pred], lr=1e-3, momentum=0.9, weight_decay=1e-4),
                        domain_loss=cdan, optimizer_d=FlatSGD([disc], lr=1e-4, momentum=0.9, weight_decay=1e-4), seed=9)
        out = tr.train_step(xs, ys, xr, None)
        runs.append((SEDTrainer.loss_value(out), float(out["domain"]), tr.arena.flat.clone()))
        assert tr.arena.flat.numel() == crnn.flat.numel() + pred.flat.numel() + disc.flat.numel()
        assert float(disc.flat_grad.abs().max()) > 0 and bool(torch.isfinite(tr.arena.flat).all())
    assert runs[0][0] == runs[1][0] and torch.equal(runs[0][2], runs[1][2])
    assert 0.3 < runs[0][1] < 1.5                     # BCE of an untrained discriminator: around log 2
