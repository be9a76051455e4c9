"""Clip_Discriminator + gradient-reverse + BCE domain loss on the GPU vs the vectors produced by the reference's
own Clip_Discriminator / ConditionalDomainAdversarialLoss (tests/golden/clipd.npz) and the torch oracle."""
import os

import numpy as np
import pytest
import torch

from oracle import crnn_oracle as co
from oracle import seeded

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("mode", ["fp32", "bf16x3"])
def test_domain_loss_and_gradients_match_reference(golden_dir, mode):
    """fp32 (every contraction on the fp32 matrix cores) reproduces the reference's vectors to rounding.  The default
    split-fp32 mode perturbs the GEMM outputs by ~1e-6 relative; at this golden's size (2 + 2 clips of 64 frames) the
    deeper BatchNorms normalise over a handful of samples per channel and their backward amplifies that ~1000x
    (measured: feature gradients 1.6e-3, last-layer parameter-gradient norms up to 4e-2), so there the gradient
    checks use tolerances that reflect the conditioning of the problem, not of the kernels; loss and eval output
    (running statistics) keep the strict ones."""
    from bsed_amd.disc import Clip_Discriminator, ConditionalDomainAdversarialLoss
    tol_g, tol_n = (2e-4, 5e-4) if mode == "fp32" else (5e-3, 8e-2)
    g = np.load(os.path.join(golden_dir, "clipd.npz"))
    B, T, seed = (int(v) for v in g["meta"])
    rng = np.random.default_rng(seed)
    f_s = rng.standard_normal((B, T, 256)).astype(np.float32)
    f_t = rng.standard_normal((B, T, 256)).astype(np.float32)
    odisc = co.Clip_Discriminator()
    seeded.load_seeded(odisc, seed + 1)
    disc = Clip_Discriminator(input_dim=8192, dropout=0.5)
    assert set(disc.state_dict().keys()) == set(odisc.state_dict().keys())
    disc.load_state_dict(odisc.state_dict())
    disc.conv_mode = mode
    disc.train()
    cdan = ConditionalDomainAdversarialLoss(disc, entropy_conditioning=False, num_classes=20, features_dim=256)
    fs, ft = torch.from_numpy(f_s).cuda(), torch.from_numpy(f_t).cuda()
    names = [str(n) for n in g["dnames"]]
    for it in range(3):
        disc.zero_grad()
        loss = cdan(None, fs, None, ft)
        dfs, dft = cdan.backward_features()
        assert abs(float(loss) - float(g[f"loss{it}"])) < 2e-5, (it, float(loss), float(g[f"loss{it}"]))
        ref = g[f"dfs{it}"]
        got = dfs.cpu().numpy()[:, ::16, ::8]
        assert np.abs(got - ref).max() <= tol_g * np.abs(ref).max() + 1e-9, it
        assert abs(float(dfs.norm()) - float(g[f"dfs_norm{it}"])) <= tol_g * float(g[f"dfs_norm{it}"]) + 1e-9
        assert abs(float(dft.norm()) - float(g[f"dft_norm{it}"])) <= tol_g * float(g[f"dft_norm{it}"]) + 1e-9
        norms = np.array([float(disc.P(n).grad.double().norm()) for n in names])
        keep = np.array([not (n.startswith("conv_") and n.endswith("bias")) for n in names])  # zero grad under BN
        np.testing.assert_allclose(norms[keep], g[f"dgrad_norms{it}"][keep], rtol=tol_n, atol=1e-8)
    disc.eval()
    with torch.no_grad():
        out = disc(torch.cat([fs, ft]))
    np.testing.assert_allclose(out.cpu().numpy(), g["eval_out"], atol=2e-5)


def test_discriminator_parameter_gradients_vs_oracle():
    from bsed_amd.disc import Clip_Discriminator
    seed, B, T = 4, 3, 101
    rng = np.random.default_rng(seed)
    f = rng.standard_normal((2 * B, T, 256)).astype(np.float32)
    odisc = co.Clip_Discriminator()
    seeded.load_seeded(odisc, seed)
    odisc.train()
    x = torch.from_numpy(f).requires_grad_()
    loss_ref = co.domain_loss(odisc, x[:B], x[B:], 0.37)
    loss_ref.backward()
    disc = Clip_Discriminator()
    disc.conv_mode = "fp32"   # wiring check at a tiny batch: see test_domain_loss_and_gradients_match_reference
    disc.load_state_dict({k: v for k, v in odisc.state_dict().items() if "num_batches" not in k or True})
    disc.nbt.zero_()
    disc.train(); disc.zero_grad()
    d, ctx = disc.run_forward(torch.from_numpy(f).cuda(), n_source=B)
    df = disc.run_backward(ctx, 0.37)
    loss = float(ctx["lossp"][:, 0, 0].sum() / (2 * B))
    assert abs(loss - float(loss_ref)) < 2e-5
    np.testing.assert_allclose(df.cpu().numpy(), x.grad.numpy(), atol=2e-4 * float(x.grad.abs().max()))
    for k, p in odisc.named_parameters():
        if k.startswith("conv_") and k.endswith("bias"):
            continue
        got, ref = disc.P(k).grad.cpu().double(), p.grad.double()
        assert float((got - ref).norm()) <= 5e-4 * float(ref.norm()) + 1e-8, k


def test_adversarial_train_step_gradients_match_oracle():
    """class losses + domain loss through GRL in one step (reference src/main_scmt_ada_weak.py:312-345,527-528)"""
    from bsed_amd.disc import Clip_Discriminator, ConditionalDomainAdversarialLoss
    from bsed_amd.engine import FlatSGD, SEDTrainer
    from bsed_amd.models import CRNN, Predictor
    seed, B, T = 17, 2, 256
    kw = dict(co.CRNN_KWARGS); kw["dropout"] = 0.0
    ocrnn, opred, odisc = co.CRNN(**kw), co.Predictor(**co.PREDICTOR_KWARGS), co.Clip_Discriminator()
    seeded.load_seeded(ocrnn, seed); seeded.load_seeded(opred, seed + 1); seeded.load_seeded(odisc, seed + 2)
    xs = seeded.db_like_input(seed + 3, B, T); xr = seeded.db_like_input(seed + 4, B, T)
    y = seeded.strong_targets(seed + 5, B, T // 4)
    for m in (ocrnn, opred, odisc):
        m.train()
    coeff = co.grl_coeff(7)
    loss_c, outs = co.train_losses(ocrnn, opred, torch.from_numpy(xs), torch.from_numpy(y), torch.from_numpy(xr))
    loss_d = co.domain_loss(odisc, outs["enc_syn"], outs["enc_real"], coeff)
    (loss_c + loss_d).backward()

    crnn, pred, disc = CRNN(**kw), Predictor(**co.PREDICTOR_KWARGS), Clip_Discriminator()
    # exact-fp32 contractions here: with B = 2 and 64 frames the discriminator's deeper BatchNorms see a handful of
    # samples per channel and amplify 1e-5 input perturbations ~100x, which would test conditioning, not wiring
    crnn.conv_mode = "fp32"
    disc.conv_mode = "fp32"
    ocrnn2 = co.CRNN(**kw); seeded.load_seeded(ocrnn2, seed)          # fresh running stats
    odisc2 = co.Clip_Discriminator(); seeded.load_seeded(odisc2, seed + 2)
    crnn.load_state_dict(ocrnn2.state_dict()); pred.load_state_dict(opred.state_dict())
    disc.load_state_dict(odisc2.state_dict())
    cdan = ConditionalDomainAdversarialLoss(disc)
    cdan.iter_num = 7
    # lr = 0 optimizers: the step leaves parameters alone so the gradients can be inspected afterwards
    tr = SEDTrainer(crnn, pred, optimizer=FlatSGD([crnn, pred], lr=0.0, momentum=0.0, weight_decay=0.0),
                    domain_loss=cdan, optimizer_d=FlatSGD([disc], lr=0.0, momentum=0.0, weight_decay=0.0))
    out = tr.train_step(torch.from_numpy(xs).cuda(), torch.from_numpy(y).cuda(), torch.from_numpy(xr).cuda())
    loss = SEDTrainer.loss_value(out)
    assert abs(loss - float(loss_c + loss_d)) < 3e-5 * abs(loss), (loss, float(loss_c + loss_d))
    bad = []
    for mod, omod in ((crnn, ocrnn), (pred, opred), (disc, odisc)):
        for k, p in omod.named_parameters():
            key = k.replace("cnn.cnn.", "cnn.", 1)
            if (".conv" in key or key.startswith("conv_")) and key.endswith("bias"):
                continue
            got, ref = mod.P(key).grad.cpu().double(), p.grad.double()
            err = float((got - ref).norm())
            # BatchNorm bias gradients are cancellation-heavy sums over ~1e5 positions: fp32 summation order alone
            # moves them by ~1e-4 absolute, hence the absolute floor
            if err > 5e-4 * float(ref.norm()) + 2e-4:
                bad.append((key, err, float(ref.norm())))
    assert not bad, bad


@pytest.mark.parametrize("crnn_mode", ["bf16x3", "fp32"])
def test_adversarial_step_in_bench_mode_vs_fp64_oracle(crnn_mode):
    """The adversarial step in the configuration ``bench.py --mode ada`` runs -- split-fp32 ("bf16x3") CRNN contractions,
    fp32-core discriminator -- at a well-conditioned size: 12 synthetic + 12 real clips of 865 frames (216 output
    frames, the BASELINE shape), against the oracle evaluated in float64.  Reference: src/models/CRNN_GRL.py:16-53,
    src/DA/cdan_frame.py:89-119, src/main_scmt_ada_weak.py:312-345,527-528.

    The bar is DERIVED in the test, not asserted: the same step is also run by the stock-torch oracle in float32 (what the
    reference itself computes), and its per-tensor relative-L2 error against the float64 oracle, e32[k], is the yardstick:
      * exact-fp32 CRNN contractions: every tensor <= 2 x max(e32[k], median of e32 over the module's tensors)
        (the median floor keeps a tensor on which the float32 oracle happened to flip few LeakyReLU mask elements from
        setting an unattainable bar; 2e-4, the gradient bar of the whole suite, is the floor: the Predictor's tensors
        receive no domain gradient and the float32 oracle is within 2e-6 of float64 on them);
      * split-fp32 CRNN contractions (the bench's mode): the same bar times sqrt(f_hip / f_32), where f_hip and f_32 are
        the MEASURED forward errors of the two float32 paths on the encoder output -- a forward error eps flips ~eps of
        the discriminator's mask elements, each flip moves a gradient element by 0.8x, so gradient error grows like
        sqrt(eps).  If this mode met only the scaled bar and not the plain 2 x e32 one, the line
        printed below says so: that is a statement about configs[4]'s default mode, not a reason to widen anything.
    Loss: 2e-5 relative in both modes."""
    from bsed_amd.disc import Clip_Discriminator, ConditionalDomainAdversarialLoss
    from bsed_amd.engine import FlatSGD, SEDTrainer
    from bsed_amd.models import CRNN, Predictor
    seed, B, T = 29, 12, 865
    kw = dict(co.CRNN_KWARGS); kw["dropout"] = 0.0
    xs = seeded.db_like_input(seed + 3, B, T); xr = seeded.db_like_input(seed + 4, B, T)
    y = seeded.strong_targets(seed + 5, B, T // 4)
    it = 700
    coeff = co.grl_coeff(it)

    def oracle_step(dtype):
        ocrnn, opred, odisc = co.CRNN(**kw), co.Predictor(**co.PREDICTOR_KWARGS), co.Clip_Discriminator()
        seeded.load_seeded(ocrnn, seed); seeded.load_seeded(opred, seed + 1); seeded.load_seeded(odisc, seed + 2)
        sd = ({k: v.clone() for k, v in ocrnn.state_dict().items()}, {k: v.clone() for k, v in opred.state_dict().items()},
              {k: v.clone() for k, v in odisc.state_dict().items()})
        for m in (ocrnn, opred, odisc):
            m.to(dtype).train()
        loss_c, outs = co.train_losses(ocrnn, opred, torch.from_numpy(xs).to(dtype), torch.from_numpy(y).to(dtype),
                                       torch.from_numpy(xr).to(dtype))
        loss_d = co.domain_loss(odisc, outs["enc_syn"], outs["enc_real"], coeff)
        (loss_c + loss_d).backward()
        grads = {}
        for tag, omod in (("crnn", ocrnn), ("pred", opred), ("disc", odisc)):
            for k, p in omod.named_parameters():
                key = k.replace("cnn.cnn.", "cnn.", 1)
                if (".conv" in key or key.startswith("conv_")) and key.endswith("bias"):
                    continue  # exactly zero under train-mode BatchNorm
                grads[(tag, key)] = p.grad.double()
        enc = torch.cat([outs["enc_syn"], outs["enc_real"]]).detach().double()
        return float(loss_c + loss_d), grads, enc, sd

    loss64, g64, enc64, (sd_c, sd_p, sd_d) = oracle_step(torch.float64)
    loss32, g32, enc32, _ = oracle_step(torch.float32)
    e32 = {k: float((g32[k] - g64[k]).norm() / g64[k].norm()) for k in g64}
    f32 = float((enc32 - enc64).abs().max())

    crnn, pred, disc = CRNN(**kw), Predictor(**co.PREDICTOR_KWARGS), Clip_Discriminator()
    assert crnn.conv_mode == "bf16x3" and disc.conv_mode == "fp32"      # what bench.py --mode ada runs
    crnn.conv_mode = crnn_mode
    crnn.load_state_dict(sd_c); pred.load_state_dict(sd_p); disc.load_state_dict(sd_d)
    crnn.train()
    with torch.no_grad():
        enc_hip = torch.cat([crnn.run_forward(torch.from_numpy(v).cuda(), save=False)[0] for v in (xs, xr)]).double().cpu()
    f_hip = float((enc_hip - enc64).abs().max())
    crnn.load_state_dict(sd_c)                                            # fresh running statistics for the step below
    cdan = ConditionalDomainAdversarialLoss(disc)
    cdan.iter_num = it
    tr = SEDTrainer(crnn, pred, optimizer=FlatSGD([crnn, pred], lr=0.0, momentum=0.0, weight_decay=0.0),
                    domain_loss=cdan, optimizer_d=FlatSGD([disc], lr=0.0, momentum=0.0, weight_decay=0.0))
    out = tr.train_step(torch.from_numpy(xs).cuda(), torch.from_numpy(y).cuda(), torch.from_numpy(xr).cuda())
    loss = SEDTrainer.loss_value(out)
    assert abs(loss - loss64) < 2e-5 * abs(loss), (loss, loss64)
    mods = {"crnn": crnn, "pred": pred, "disc": disc}
    med = {tag: float(np.median([v for (t_, _), v in e32.items() if t_ == tag])) for tag in mods}
    scale = 1.0 if crnn_mode == "fp32" else max(1.0, (f_hip / f32) ** 0.5)
    rows, bad, over_plain = [], [], 0
    for (tag, key), ref in g64.items():
        got = mods[tag].P(key).grad.cpu().double()
        err = float((got - ref).norm() / ref.norm())
        base = max(2.0 * max(e32[(tag, key)], med[tag]), 2e-4)     # never tighter than the suite-wide gradient bar
        rows.append((key, err, e32[(tag, key)]))
        over_plain += err > base
        if err > scale * base:
            bad.append((key, err, e32[(tag, key)], scale * base))
    rows.sort(key=lambda r: -r[1] / max(r[2], 1e-12))
    print(f"adversarial step, CRNN {crnn_mode} + fp32 discriminator vs fp64 oracle: forward error on the encoding "
          f"{f_hip:.2e} (float32 torch oracle {f32:.2e}), bar scale {scale:.2f}; per-tensor relative L2 "
          f"median HIP {np.median([r[1] for r in rows]):.2e} / float32 oracle {np.median([r[2] for r in rows]):.2e}; "
          f"largest ratios: {[(k, f'{e:.1e}', f'{r:.1e}') for k, e, r in rows[:5]]}; tensors above the PLAIN 2 x e32 bar: "
          f"{over_plain} of {len(rows)}")
    assert not bad, bad
    if crnn_mode == "fp32":
        assert over_plain == 0
