"""Frame_Discriminator (reference src/models/CRNN_GRL.py:116-140) on the HIP path against vectors produced by the
reference's own module (tests/golden/frame_d.npz): forward output, input gradient and every parameter gradient for a
fixed upstream gradient, in both contraction modes; dropout masks shared by forward and backward; the reference's error
behaviour when it is paired with the clip-level domain loss."""
import os

import numpy as np
import pytest
import torch

from oracle import seeded

pytestmark = pytest.mark.gpu


def _build(seed, dropout, mode):
    from bsed_amd.disc import Frame_Discriminator
    m = Frame_Discriminator(input_dim=256, dropout=dropout)
    vals = seeded.seeded_state({k: tuple(v.shape) for k, v in m.state_dict().items()}, seed)
    m.load_state_dict({k: torch.from_numpy(v) for k, v in vals.items()})
    m.conv_mode = mode
    return m, vals


@pytest.mark.parametrize("mode", ["fp32", "bf16x3"])
def test_frame_discriminator_matches_reference(golden_dir, mode):
    g = np.load(os.path.join(golden_dir, "frame_d.npz"))
    N, T, seed = (int(v) for v in g["meta"])
    x = torch.from_numpy(np.random.default_rng(seed).standard_normal((N, T, 256)).astype(np.float32)).cuda()
    m, vals = _build(seed + 1, 0.0, mode)
    assert sorted(m.state_dict().keys()) == sorted(str(n) for n in g["state_names"])
    assert seeded.checksum(vals) == float(g["weight_checksum"][0])
    m.train(); m.zero_grad()
    d, ctx = m.run_forward(x)
    assert d.shape == (N, T, 1)
    assert float(np.abs(d.cpu().numpy() - g["out"]).max()) < 2e-6
    up = (torch.cos(torch.arange(d.numel(), dtype=torch.float32)).view_as(d) * 0.3).cuda()
    dx = m.run_backward(ctx, up)
    tol = 2e-4 if mode == "fp32" else 5e-3        # LeakyReLU kinks (measured 2.4e-3 on dense_d_1.bias): see the Clip_Discriminator tests
    ref = g["dx"]
    assert np.abs(dx.cpu().numpy()[:, ::7, ::5] - ref).max() <= tol * np.abs(ref).max()
    assert abs(float(dx.norm()) - float(g["dx_norm"])) <= tol * float(g["dx_norm"])
    for k, p in m.named_parameters():
        r = g["grad/" + k]
        err = float(np.linalg.norm(p.grad.cpu().numpy().astype(np.float64) - r) / (np.linalg.norm(r) + 1e-30))
        assert err <= tol, (k, err)
    # the autograd bridge gives the same numbers
    m.zero_grad()
    xr = x.clone().requires_grad_()
    (m(xr) * up).sum().backward()
    assert float((xr.grad - dx).abs().max()) == 0.0
    # eval: no dropout, no graph
    m.eval()
    with torch.no_grad():
        assert float((m(x) - d).abs().max()) < 1e-6


def test_frame_discriminator_dropout_masks_and_error_behaviour():
    from bsed_amd.disc import ConditionalDomainAdversarialLoss
    m, _ = _build(5, 0.5, "fp32")
    m.train(); m.set_seed(77)
    x = torch.randn(4, 64, 256, device="cuda", generator=torch.Generator(device="cuda").manual_seed(1))
    d1, ctx = m.run_forward(x)
    d2, _ = m.run_forward(x)
    assert torch.equal(d1, d2)                       # stateless counter RNG: same seed, same masks
    m.set_seed(78)
    d3, _ = m.run_forward(x)
    assert not torch.equal(d1, d3)
    # backward is linear in the upstream gradient with the forward's masks
    m.set_seed(77)
    ups = [torch.randn(d1.shape, device="cuda", generator=torch.Generator(device="cuda").manual_seed(s)) for s in (2, 3)]
    outs = []
    for up in (ups[0], ups[1], 0.5 * ups[0] - 2.0 * ups[1]):
        _, c = m.run_forward(x)
        m.zero_grad()
        dx = m.run_backward(c, up)
        outs.append((dx.clone(), m.flat_grad.clone()))
    for i in (0, 1):
        lin = 0.5 * outs[0][i] - 2.0 * outs[1][i]
        assert float((outs[2][i] - lin).norm() / lin.norm()) < 1e-4
    # half of the hidden activations are dropped
    h1 = c["h1"]
    assert 0.45 < float((h1 == 0).float().mean()) < 0.55
    with pytest.raises(ValueError):
        ConditionalDomainAdversarialLoss(m)          # (N,T,1) scores vs per-clip labels: the reference raises too
