"""The first CNN block WITHOUT its conv output / gradient tensors in HBM (csrc/block0.hip) against the four-kernel
form that stores them (conv0_fwd, glu16_fwd, glu16_bwd, conv0_wgrad), against the oracle's conv0 output, and -- through
the whole network -- against the oracle's gradients (reference: src/models/CNN.py:46-67 with i = 0).

Bars: in eval mode (BatchNorm from the running statistics) the block's output is BIT-IDENTICAL to the four-kernel form
(same FMA chain, same dropout counters); in train mode the batch statistics agree to summation order (2e-6 relative:
the block's output then differs in the last bit), so with exact-fp32 contractions behind it the network output agrees
to 1e-5 and every parameter gradient to 1e-5 relative L2 (measured 2.6e-6 / 3.7e-6; conv0's weight gradient is
assembled analytically from Gx / R / Sx in fp64 instead of being summed position by position in fp32).  (With the
default split-fp32 contractions a last-bit input change moves each layer by the mode's own ~2e-6 rounding noise: the
two forms then sit 1.2e-5 apart and equally far from the oracle -- tools/block0_diag.py.)  Against the oracle the
block's gradients meet the bar of tests/test_crnn_gpu.py (2e-4) in both modes.
"""
import numpy as np
import pytest
import torch

from oracle import crnn_oracle as co
from oracle import seeded

pytestmark = pytest.mark.gpu


def _pair(dropout, seed, conv_mode="bf16x3"):
    from bsed_amd.models import CRNN
    kw = dict(co.CRNN_KWARGS)
    kw["dropout"] = dropout
    ocrnn = co.CRNN(**kw)
    seeded.load_seeded(ocrnn, seed)
    # a BatchNorm that is not the identity, so that every coefficient of its backward matters
    with torch.no_grad():
        bn0 = ocrnn.cnn.cnn.batchnorm0
        bn0.weight.copy_(torch.linspace(0.5, 1.5, 16))
        bn0.bias.copy_(torch.linspace(-0.3, 0.3, 16))
        ocrnn.cnn.cnn.conv0.bias.copy_(torch.linspace(-0.2, 0.2, 16))
    mods = []
    for fused in (True, False):
        m = CRNN(**kw)
        m.conv_mode = conv_mode
        m.block0_fused = fused
        m.load_state_dict(ocrnn.state_dict())
        mods.append(m)
    return ocrnn, mods[0], mods[1]


BLOCK0 = ("cnn.conv0.weight", "cnn.batchnorm0.weight", "cnn.batchnorm0.bias", "cnn.glu0.linear.weight",
          "cnn.glu0.linear.bias")


@pytest.mark.parametrize("B,T,dropout", [(3, 65, 0.5), (2, 64, 0.0), (5, 217, 0.5)])
def test_fused_block0_equals_the_four_kernel_form(B, T, dropout):
    """T = 65 / 217: an odd number of frames, the last row is outside the pooled extent (zero gradient there, but the row
    still counts in the batch statistics and in the BatchNorm-backward mean terms)"""
    seed = 31
    x = torch.from_numpy(seeded.db_like_input(seed, B, T)).cuda()
    _, fused, plain = _pair(dropout, seed, "fp32")
    outs = []
    for m in (fused, plain):
        m.train()
        m.set_seed(77)
        enc, ctx = m.run_forward(x, save=True)
        blk = ctx["blocks"][0]
        assert (blk["y"] is None) == (m is fused)
        d = torch.from_numpy(np.random.default_rng(5).standard_normal(tuple(enc.shape)).astype(np.float32)).cuda()
        m.zero_grad()
        m.run_backward(ctx, d * 1e-2)
        outs.append((enc, ctx, blk))
    (ef, cf, bf), (ep, cp, bp) = outs
    np.testing.assert_allclose(cf["blocks"][1]["inp"].cpu().numpy(), cp["blocks"][1]["inp"].cpu().numpy(),
                               rtol=1e-6, atol=3e-6, err_msg="pooled output of block 0")
    np.testing.assert_allclose(ef.cpu().numpy(), ep.cpu().numpy(), rtol=0, atol=1e-5)
    for k in ("mean", "invstd", "scale", "shift"):
        np.testing.assert_allclose(bf[k].cpu().numpy(), bp[k].cpu().numpy(), rtol=2e-6, atol=1e-7, err_msg=k)
    for k in ("running_mean", "running_var"):
        np.testing.assert_allclose(fused.P("cnn.batchnorm0." + k).cpu().numpy(),
                                   plain.P("cnn.batchnorm0." + k).cpu().numpy(), rtol=2e-6, atol=1e-7)
    assert int(fused.P("cnn.batchnorm0.num_batches_tracked")) == 1
    for name, p in fused.named_parameters():
        if ".conv" in name and name.endswith(".bias"):
            continue   # exactly zero under train-mode BatchNorm
        a, b = p.grad.double(), plain.P(name).grad.double()
        err = float((a - b).norm())
        assert err <= 1e-5 * float(b.norm()) + 1e-9, (name, err, float(b.norm()))


def test_fused_block0_kernels_on_other_map_widths():
    """the block's kernels on maps that are not 128 wide (40: partial 16-column wave tiles; 192: two column chunks in
    the backward kernel, three waves per row in the statistics kernel) against the four-kernel form, kernel by kernel"""
    from bsed_amd import ops
    g = torch.Generator(device="cuda").manual_seed(11)
    for B, H, W in ((3, 33, 40), (2, 18, 192)):
        x = torch.rand(B, H, W, device="cuda", generator=g) * 60 - 70
        cw = torch.randn(16, 1, 3, 3, device="cuda", generator=g) * 0.3
        cb = torch.randn(16, device="cuda", generator=g) * 0.1
        wg = torch.randn(16, 16, device="cuda", generator=g) * 0.2
        bg = torch.randn(16, device="cuda", generator=g) * 0.1
        y, st = ops.conv0_fwd(x, cw, cb, B, H, W, 16, want_stats=True)
        st_f, xr64 = ops.block0_stats(x, cw, cb, B, H, W)
        np.testing.assert_allclose(st_f.double().sum(0).cpu().numpy(), st.double().sum(0).cpu().numpy(), rtol=1e-5, atol=1e-3)
        # Sx / R against a direct evaluation
        xp = torch.nn.functional.pad(x.double(), (1, 1, 1, 1))
        taps = torch.stack([xp[:, kh:kh + H, kw:kw + W] for kh in range(3) for kw in range(3)], 0).reshape(9, -1)
        want = torch.cat([taps.sum(1), torch.stack([(taps[t] * taps[u]).sum() for t in range(9) for u in range(t, 9)])])
        np.testing.assert_allclose(xr64.cpu().numpy(), want.cpu().numpy(), rtol=1e-5)
        scale = torch.rand(16, device="cuda", generator=g) * 0.05 + 0.02
        shift = torch.randn(16, device="cuda", generator=g) * 0.1
        for pool in ((2, 2), (1, 2)):
            a = ops.block0_fwd(x, cw, cb, scale, shift, wg, bg, B, H, W, pool, 0.5, 100, 7)
            b = ops.glu16_fwd(y, scale, shift, wg, bg, B, H, W, pool, 0.5, 100, 7)
            assert torch.equal(a, b), (W, pool)
            dp = torch.randn(tuple(a.shape), device="cuda", generator=g) * 1e-2
            pdw, pdb, pst, pgx, G = ops.block0_bwd(x, cw, cb, scale, shift, wg, bg, dp, B, H, W, pool, 0.5, 100, 7)
            gq, qdw, qdb, qst, G2 = ops.glu16_bwd(y, scale, shift, wg, bg, dp, B, H, W, pool, 0.5, 100, 7)
            for u, v, name in ((pdw, qdw, "dw"), (pdb[:, 0], qdb[:, 0], "db"), (pst, qst, "st")):
                np.testing.assert_allclose(u.double().sum(0).cpu().numpy(), v.double().sum(0).cpu().numpy(),
                                           rtol=2e-4, atol=1e-6, err_msg=f"{name} W={W} pool={pool}")
            gx = torch.einsum("bhwc,tbhw->tc", gq.double(), taps.view(9, B, H, W))     # sum g x_tap, (9, 16)
            np.testing.assert_allclose(pgx.double().sum(0).cpu().numpy(), gx.cpu().numpy(), rtol=2e-4, atol=1e-6)


def test_fused_block0_eval_mode_is_bit_identical():
    seed, B, T = 8, 2, 64
    x = torch.from_numpy(seeded.db_like_input(seed, B, T)).cuda()
    _, fused, plain = _pair(0.5, seed)
    with torch.no_grad():
        for m in (fused, plain):   # running statistics that are not the initial (0, 1)
            m.P("cnn.batchnorm0.running_mean").copy_(torch.linspace(-1, 1, 16))
            m.P("cnn.batchnorm0.running_var").copy_(torch.linspace(0.5, 2, 16))
            m.eval()
    a, _ = fused.run_forward(x, save=False)
    b, _ = plain.run_forward(x, save=False)
    assert torch.equal(a, b)


@pytest.mark.parametrize("conv_mode", ["bf16x3", "fp32"])
def test_fused_block0_gradients_vs_oracle(conv_mode):
    """the oracle's autograd through the whole CRNN: block-0 gradients of the fused form inside the 2e-4 bar"""
    seed, B, T = 12, 3, 65
    x = torch.from_numpy(seeded.db_like_input(seed, B, T))
    ocrnn, fused, _ = _pair(0.0, seed, conv_mode)
    ocrnn.train(); fused.train()
    enc_ref, _ = ocrnn(x)
    d = torch.from_numpy(np.random.default_rng(6).standard_normal(tuple(enc_ref.shape)).astype(np.float32)) * 1e-2
    ocrnn.zero_grad()
    enc_ref.backward(d)
    enc, ctx = fused.run_forward(x.cuda(), save=True)
    assert float((enc.cpu() - enc_ref.detach()).abs().max()) < 1e-4
    fused.zero_grad()
    fused.run_backward(ctx, d.cuda())
    ref = {k.replace("cnn.cnn.", "cnn."): p for k, p in ocrnn.named_parameters()}
    for name in BLOCK0:
        a, b = fused.P(name).grad.cpu().double(), ref[name].grad.double()
        err = float((a - b).norm())
        assert err <= 2e-4 * float(b.norm()) + 1e-7, (name, err, float(b.norm()))
    np.testing.assert_allclose(fused.P("cnn.batchnorm0.running_var").cpu().numpy(),
                               ocrnn.cnn.cnn.batchnorm0.running_var.numpy(), rtol=2e-4)
    np.testing.assert_allclose(fused.P("cnn.batchnorm0.running_mean").cpu().numpy(),
                               ocrnn.cnn.cnn.batchnorm0.running_mean.numpy(), rtol=2e-4, atol=1e-6)


@pytest.mark.parametrize("B,H,pool", [(2, 64, (2, 2)), (3, 65, (2, 2)), (2, 40, (1, 2))])
def test_block0_bf16_instances_match_their_fp32_twins(B, H, pool):
    """bf16 mode (conv_mode = "bf16"): b0_fwd / b0_bwd<.., 1> take / return bf16 pooled tensors and run their four
    16 x 16 x 16 contractions per position as single v_mfma_f32_16x16x16_bf16 (operands rounded to bf16, fp32
    accumulation) -- against the fp32 instances (exact-fp32 v_mfma_f32_16x16x4_f32) on the same inputs, same dropout
    masks.  Bars: forward 1e-2 relative L2 (three chained bf16 roundings of 2^-9 each; measured 3.3e-3 .. 4.1e-3), every partial
    sum of the backward pass 2e-2 of its fp32 twin's norm (measured <= 5.8e-3)."""
    from bsed_amd import ops
    W, C = 128, 16
    rng = np.random.default_rng(5 * H + B)
    x = torch.from_numpy(seeded.db_like_input(9, B, H)).cuda().reshape(B, H, W).contiguous()
    cw = torch.from_numpy((rng.standard_normal((C, 9)) * 0.3).astype(np.float32)).cuda()
    cb = torch.from_numpy((rng.standard_normal(C) * 0.1).astype(np.float32)).cuda()
    scale = torch.from_numpy(rng.uniform(0.02, 0.06, C).astype(np.float32)).cuda()     # dB-scale inputs: y ~ 20
    shift = torch.from_numpy((rng.standard_normal(C) * 0.3).astype(np.float32)).cuda()
    wg = torch.from_numpy((rng.standard_normal((C, C)) * 0.3).astype(np.float32)).cuda()
    bg = torch.from_numpy((rng.standard_normal(C) * 0.1).astype(np.float32)).cuda()
    ph, pw = pool
    outs = {dt: ops.block0_fwd(x, cw, cb, scale, shift, wg, bg, B, H, W, pool, 0.5, 100, 4242, out_dtype=dt)
            for dt in (torch.float32, torch.bfloat16)}
    ref, got = outs[torch.float32], outs[torch.bfloat16].float()
    assert float((got - ref).norm() / ref.norm()) < 1e-2
    assert float(ref.abs().max()) > 0.1            # the comparison is not between two zero tensors
    dpool = torch.from_numpy(rng.standard_normal((B, H // ph, W // pw, C)).astype(np.float32)).cuda() * 1e-2
    parts = {}
    for dt in (torch.float32, torch.bfloat16):
        parts[dt] = ops.block0_bwd(x, cw, cb, scale, shift, wg, bg, dpool.to(dt), B, H, W, pool, 0.5, 100, 4242)
    # the fp32 twin is fed the SAME (bf16-rounded) upstream gradient, so only the contraction arithmetic differs
    parts[torch.float32] = ops.block0_bwd(x, cw, cb, scale, shift, wg, bg, dpool.to(torch.bfloat16).float(), B, H, W,
                                          pool, 0.5, 100, 4242)
    for name, a, b in zip(("dW_glu", "db_glu", "bn sums", "Gx"), parts[torch.bfloat16][:4], parts[torch.float32][:4]):
        a, b = a.double().sum(0), b.double().sum(0)
        assert float((a - b).norm()) < 2e-2 * float(b.norm()) + 1e-9, (name, float((a - b).norm() / b.norm()))
