"""The N-split split-fp32 convolution kernel (csrc/igemm3n.hip, round 3) against the slab kernel of rounds 1-2
(csrc/igemm3.hip) and against a float64 reference: the layers are the reference's seven nn.Conv2d
(/root/reference/src/models/CNN.py:46-47), the GRU input projections (src/models/RNN.py:7-16) and the
discriminator's shapes.  With 32 x 32 x 16 MFMAs the accumulation order per output element is the slab kernel's => the
output tensors are BIT-identical; with 16 x 16 x 32 MFMAs (the default for fp32 activations: faster) a 32-channel chunk
is summed inside ONE instruction instead of two, and the outputs agree to rounding (bar: 4e-6 of max|ref|, measured
<= 1.5e-6).  The BatchNorm partial sums are summed in a different (fixed) order and compared in float64."""
import os

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

T3 = [(kh - 1, kw - 1) for kh in range(3) for kw in range(3)]
FL = [(-a, -b) for a, b in T3]
# (NB, H, W, CIN, N, taps, stats, valid)
SHAPES = [
    (2, 21, 16, 64, 128, T3, True, None),       # BN = 128, two chunks, partial last tile row
    (1, 40, 8, 128, 128, T3, True, None),       # the 216 x 8 layer's shape class (PV = 6, WPE = 3 build)
    (2, 70, 4, 128, 128, T3, True, None),       # PV = 9 (34 x 6 patch)
    (3, 130, 2, 128, 128, FL, False, None),     # 66 x 4 patch, flipped taps (data gradient)
    (1, 37, 32, 32, 64, T3, True, None),        # BN = 64, ONE chunk (single patch buffer)
    (2, 19, 16, 128, 64, FL, False, None),      # BN = 64, four chunks
    (1, 25, 32, 64, 32, FL, False, None),       # BN = 32: the four waves share the weight fragments
    (1, 1000, 1, 128, 768, ((0, 0),), False, None),   # GRU input projection: 1 tap, six column blocks
    (1, 700, 1, 768, 256, ((0, 0),), False, None),    # its data gradient: 24 chunks x 1 tap
    (2, 313, 1, 128, 128, ((-1, 0), (0, 0), (1, 0)), True, None),   # FPN level: 3 x 1 stencil on a width-1 map
    (2, 24, 16, 64, 64, ((0, 0), (0, 1), (1, 0), (1, 1)), True, (23, 15)),   # discriminator: 2 x 2 taps, valid extent
    (1, 33, 8, 32, 20, T3, True, None),         # N not a multiple of 32 (masked channels)
]


def _run(nsplit, x, w, bias, NB, H, W, CIN, N, taps, stats, valid, knob=0, shape=0):
    from bsed_amd import ops
    os.environ["BSED_IGEMM3N"] = "1" if nsplit else "0"
    try:
        if nsplit:
            ops.set_igemm3n_wpe(knob)
            ops.set_igemm3n_shape(shape)
        w3 = ops.pack_weight3(w, len(taps), CIN, N, CIN * N, N, 1)
        assert (w3.dim() == 6) == nsplit
        return ops.igemm3(x, w3, N, NB, H, W, CIN, taps, bias=bias, epilogue=ops.EPI_STATS if stats else ops.EPI_PLAIN,
                          valid=valid)
    finally:
        os.environ.pop("BSED_IGEMM3N", None)
        if nsplit:
            ops.set_igemm3n_wpe(0)
            ops.set_igemm3n_shape(0)


@pytest.mark.parametrize("NB,H,W,CIN,N,taps,stats,valid", SHAPES)
def test_nsplit_kernel_is_bit_identical_to_the_slab_kernel(NB, H, W, CIN, N, taps, stats, valid):
    g = torch.Generator().manual_seed(H * 131 + W)
    x = torch.randn(NB, H, W, CIN, generator=g).cuda()
    w = (torch.randn(len(taps), CIN, N, generator=g) / (len(taps) * CIN) ** 0.5).cuda()
    bias = torch.randn(N, generator=g).cuda()
    ref, ref_st = _run(False, x, w, bias, NB, H, W, CIN, N, taps, stats, valid)
    for shape in (32, 16):
        for knob in (0, 2, 3, 8):
            out, st = _run(True, x, w, bias, NB, H, W, CIN, N, taps, stats, valid, knob, shape)
            torch.cuda.synchronize()
            if shape == 32:
                assert torch.equal(out, ref), f"shape {shape} knob {knob}: max diff {float((out - ref).abs().max())}"
            else:   # (instances without a 16 x 16 x 32 form run the 32 x 32 x 16 one and pass trivially)
                err = float((out - ref).abs().max())
                assert err <= 4e-6 * float(ref.abs().max()), f"shape 16 knob {knob}: max diff {err}"
            if stats:
                a, b = st.double().sum(0).cpu().numpy(), ref_st.double().sum(0).cpu().numpy()
                np.testing.assert_allclose(a, b, rtol=2e-5, atol=2e-4 * np.sqrt(NB * H * W))


@pytest.mark.parametrize("NB,H,W,CIN,N,taps,stats,valid", SHAPES[:7])
def test_nsplit_kernel_matches_float64(NB, H, W, CIN, N, taps, stats, valid):
    """value check that does not go through the old kernel: conv in float64 on the host, 3e-5 of max|ref|; the
    statistics rows against the float64 sums of the kernel's own output"""
    g = torch.Generator().manual_seed(H * 7 + W)
    x = torch.randn(NB, H, W, CIN, generator=g)
    w = torch.randn(len(taps), CIN, N, generator=g) / (len(taps) * CIN) ** 0.5
    bias = torch.randn(N, generator=g)
    out, st = _run(True, x.cuda(), w.cuda(), bias.cuda(), NB, H, W, CIN, N, taps, stats, valid)
    xp = torch.nn.functional.pad(x.double(), (0, 0, 1, 1, 1, 1))
    ref = bias.double().expand(NB, H, W, N).clone()
    for t, (dh, dw) in enumerate(taps):
        ref += xp[:, 1 + dh:1 + dh + H, 1 + dw:1 + dw + W, :] @ w[t].double()
    err = float((out.cpu().double() - ref).abs().max())
    assert err < 3e-5 * float(ref.abs().max()), err
    if stats:
        s = st.double().sum(0).cpu()
        o = out.cpu().double().reshape(-1, N)
        np.testing.assert_allclose(s[0].numpy(), o.sum(0).numpy(), rtol=1e-5, atol=1e-3)
        np.testing.assert_allclose(s[1].numpy(), (o * o).sum(0).numpy(), rtol=1e-5, atol=1e-3)
