"""Statistics of the per-block dropout masks (the counter-hash RNG of csrc/bsed_common.h that every fused GLU kernel
regenerates in forward and backward instead of storing masks).  The reference draws torch's Bernoulli(0.5) masks
(src/models/CNN.py:59-61); parity of a random stream is statistical:

  * uniformity and independence inside a pooling window: with conv weights 0, BatchNorm bias 10 and a GLU linear of
    (W = 0, b = 1) every pre-dropout activation is sigmoid(10), so a block's pooled output is the window mean of its
    mask x 2 -- k kept elements of n = 2 or 4 follow Binomial(n, 1/2) (chi-square per block, all seven kernels paths:
    glu16, glu_fwd3<32|64|128>);
  * independence across blocks (streams), across seeds and across neighbouring windows: correlations ~ 0;
  * forward and backward regenerate the SAME mask: with an all-ones upstream gradient the GLU bias gradient equals the
    sum of the block's pooled outputs / sigmoid(10) exactly as far as fp32 summation goes."""
import math

import numpy as np
import pytest
import torch

from oracle import crnn_oracle as co

pytestmark = pytest.mark.gpu

B, T, F = 6, 128, 128
SIG = 1.0 / (1.0 + math.exp(-10.0))


def _crafted(seed):
    from bsed_amd.models import CRNN
    kw = dict(co.CRNN_KWARGS); kw["dropout"] = 0.5
    m = CRNN(**kw)
    with torch.no_grad():
        m.flat.zero_()
        for i in range(7):
            m.P(f"cnn.batchnorm{i}.weight").fill_(1.0)
            m.P(f"cnn.batchnorm{i}.bias").fill_(10.0)
            m.P(f"cnn.glu{i}.linear.bias").fill_(1.0)
    m.train(); m.set_seed(seed)
    return m


def _pooled_outputs(m, x):
    ctx = {"B": x.shape[0], "blocks": [], "train": True, "seed": m.seed, "x": x}
    a, _ = m._cnn_forward(x, ctx)
    outs = [ctx["blocks"][i + 1]["inp"] for i in range(6)] + [a]
    return outs, ctx


def test_block_masks_are_fair_independent_and_shared_by_forward_and_backward():
    x = torch.zeros((B, 1, T, F), device="cuda")
    m = _crafted(1234)
    outs, ctx = _pooled_outputs(m, x)
    kept = []
    for i, o in enumerate(outs):
        ph, pw = m.pooling[i]
        n = ph * pw
        k = (o.double() * (n / (2.0 * SIG))).cpu().numpy()          # kept elements per window
        kr = np.rint(k)
        assert np.abs(k - kr).max() < 1e-3, i                        # window means are multiples of 2 sigmoid(10) / n
        counts = np.array([(kr == j).sum() for j in range(n + 1)], dtype=np.float64)
        expect = np.array([math.comb(n, j) for j in range(n + 1)], dtype=np.float64) / 2 ** n * kr.size
        chi2 = float(((counts - expect) ** 2 / expect).sum())
        # chi-square with n degrees of freedom: 99.99 % quantiles 18.5 (n = 2) and 23.5 (n = 4)
        assert chi2 < (18.5 if n == 2 else 23.5), (i, chi2, counts, expect)
        kept.append(kr)
        # neighbouring windows along time and along channels are uncorrelated
        z = kr - kr.mean()
        for a_, b_ in ((z[:, 1:], z[:, :-1]), (z[..., 1:], z[..., :-1])):
            r = float((a_ * b_).mean() / z.var())
            assert abs(r) < 6.0 / math.sqrt(a_.size), (i, r)
    # blocks 3..6 share the time axis and the channel count: their masks (different streams) are uncorrelated
    for i in range(3, 6):
        a_, b_ = kept[i][:, :, :1, :] - kept[i].mean(), kept[i + 1][:, :, :1, :] - kept[i + 1].mean()
        if a_.shape == b_.shape:
            r = float((a_ * b_).mean() / math.sqrt(a_.var() * b_.var()))
            assert abs(r) < 6.0 / math.sqrt(a_.size), (i, r)
    # another seed: another mask; the same seed: the same mask
    outs2, _ = _pooled_outputs(_crafted(1235), x)
    outs3, _ = _pooled_outputs(_crafted(1234), x)
    for i in range(7):
        assert torch.equal(outs[i], outs3[i])
        z1, z2 = (outs[i] - outs[i].mean()).double(), (outs2[i] - outs2[i].mean()).double()
        r = float((z1 * z2).mean() / torch.sqrt(z1.var() * z2.var()))
        assert abs(r) < 6.0 / math.sqrt(z1.numel()), (i, r)
    # backward regenerates the forward's masks: d(sum of pooled outputs)/d(glu bias) = sum of pooled outputs / b, b = 1
    m.zero_grad(); m._attach_grads()
    for i in range(6, -1, -1):
        blk = ctx["blocks"][i]
        m._block_backward(blk, torch.ones_like(outs[i]), B, ctx["seed"], need_dgrad=False)
        got = m.P(f"cnn.glu{i}.linear.bias").grad.double().cpu()
        want = outs[i].double().sum(dim=(0, 1, 2)).cpu()
        assert float((got - want).abs().max()) <= 2e-5 * float(want.abs().max()), (i, got[:4], want[:4])
