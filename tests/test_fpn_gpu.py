"""CRNN_fpn (reference src/models/CRNN_GRL.py:293-389, src/models/CNN_FPN.py:33-100) on
the HIP path against the golden vectors produced by the reference's own CRNN_fpn (tests/golden/crnn_fpn.npz,
oracle/gen_golden.py::fpn_case) and against the torch CPU restatement (oracle.crnn_oracle.CRNN_fpn).

Tolerances as for the CRNN: activations 1e-4 absolute, gradients 2e-4 relative L2 per tensor."""
import os

import numpy as np
import pytest
import torch

from oracle import crnn_oracle as co
from oracle import seeded

pytestmark = pytest.mark.gpu


def _pair(dropout, seed, conv_mode):
    from bsed_amd.models import CRNN_fpn
    kw = dict(co.CRNN_KWARGS)
    kw["dropout"] = dropout
    ref = co.CRNN_fpn(**kw)
    seeded.load_seeded(ref, seed)
    mine = CRNN_fpn(**kw)
    mine.conv_mode = conv_mode
    mine.load_state_dict(ref.state_dict())
    return ref, mine


@pytest.mark.parametrize("conv_mode", ["bf16x3", "fp32"])
def test_fpn_eval_forward_matches_reference_golden(golden_dir, conv_mode):
    g = np.load(os.path.join(golden_dir, "crnn_fpn.npz"))
    B, T, seed = (int(v) for v in g["meta"])
    ref, mine = _pair(0.5, seed, conv_mode)
    # exactly the reference module's state-dict entries, "cnn.cnn.conv0.weight" next to "cnn.cnn_fcn.weight" included
    # (CNN_FPN keeps its Sequential's level; the plain CRNN's CNN strips it)
    assert sorted(mine.state_dict().keys()) == sorted(str(n) for n in g["state_names"])
    assert isinstance(mine.load_state_dict(mine.state_dict()).missing_keys, list)   # both spellings load
    mine.eval()
    x = torch.from_numpy(seeded.db_like_input(seed + 10, B, T)).cuda()
    with torch.no_grad():
        enc, d_in = mine(x)
    assert enc is d_in and enc.shape == (B, 313, 256)
    assert float(np.abs(enc.cpu().numpy() - g["eval_enc"]).max()) < 1e-4


@pytest.mark.parametrize("conv_mode", ["bf16x3", "fp32"])
def test_fpn_train_forward_backward_matches_reference_golden(golden_dir, conv_mode):
    g = np.load(os.path.join(golden_dir, "crnn_fpn.npz"))
    B, T, seed = (int(v) for v in g["meta"])
    ref, mine = _pair(0.0, seed, conv_mode)
    mine.FPN_DROPOUT = 0.0  # the golden was taken with the pyramid levels' fixed Dropout(0.5) switched off
    mine.train()
    x = torch.from_numpy(seeded.db_like_input(seed + 10, B, T)).cuda()
    enc, ctx = mine.run_forward(x, save=True)
    assert float(np.abs(enc.cpu().numpy() - g["train_enc"]).max()) < 1e-4
    w = (torch.sin(torch.arange(enc.numel(), dtype=torch.float32)).view_as(enc) * 1e-2).cuda()
    mine.zero_grad(); mine._attach_grads()
    mine.run_backward(ctx, w)
    grads = {k: p.grad for k, p in mine.named_parameters()}
    names = [str(n) for n in g["grad_names"]]
    bad = []
    for n, ref_norm in zip(names, g["grad_norms"]):
        k = n.replace("cnn.cnn.", "cnn.", 1) if n.startswith("cnn.cnn.") else n
        if ".conv" in k and k.endswith(".bias") and "conv1x1" not in k or k == "cnn.cnn_fcn.bias":
            continue  # conv bias under train-mode BatchNorm: exactly zero here, round-off in PyTorch
        mine_norm = float(grads[k].double().norm())
        if abs(mine_norm - ref_norm) > 2e-4 * ref_norm + 1e-7:
            bad.append((k, mine_norm, float(ref_norm)))
        key = "grad/" + n
        if key in g.files:
            err = float(np.linalg.norm(grads[k].cpu().numpy().astype(np.float64) - g[key])) / (float(ref_norm) + 1e-30)
            if err > 2e-4 and float(ref_norm) > 1e-6:
                bad.append((k, "tensor", err))
    assert not bad, bad
    # the unused 1x1 convolution of CNN_FPN keeps a zero gradient; the shared BatchNorm was updated twice
    assert float(grads["cnn.conv1x1.weight"].abs().max()) == 0.0
    sd = mine.state_dict()
    assert int(sd["cnn.bn_fcn.num_batches_tracked"]) == int(g["after/cnn.bn_fcn.num_batches_tracked"]) == 2
    np.testing.assert_allclose(sd["cnn.bn_fcn.running_mean"].cpu().numpy(), g["after/cnn.bn_fcn.running_mean"], atol=2e-5)
    np.testing.assert_allclose(sd["cnn.bn_fcn.running_var"].cpu().numpy(), g["after/cnn.bn_fcn.running_var"], rtol=2e-4, atol=1e-6)


def test_fpn_train_step_with_predictor_and_dropout_runs_and_is_repeatable():
    """the -fpn training configuration end to end (both dropouts on): finite loss, bitwise repeatable gradients"""
    from bsed_amd.engine import FlatAdam, SEDTrainer
    from bsed_amd.models import CRNN_fpn, Predictor, weights_init
    B, T = 4, 1255
    x = torch.from_numpy(seeded.db_like_input(3, B, T)).cuda()
    y = torch.from_numpy(seeded.strong_targets(4, B, 313)).cuda()
    grads = []
    for rep in range(2):
        torch.manual_seed(1)
        crnn, pred = CRNN_fpn(**co.CRNN_KWARGS), Predictor(**co.PREDICTOR_KWARGS)
        weights_init(crnn); weights_init(pred)
        tr = SEDTrainer(crnn, pred, optimizer=FlatAdam([crnn, pred], lr=1e-3), seed=9)
        out = tr.train_step(x, y)
        loss = SEDTrainer.loss_value(out)
        assert np.isfinite(loss) and 0.1 < loss < 5.0
        grads.append(crnn.flat_grad.clone())
    assert torch.equal(grads[0], grads[1])


@pytest.mark.parametrize("B,T_in,T_out,C", [(2, 78, 156, 256), (3, 156, 313, 256), (1, 5, 11, 8), (2, 7, 7, 4)])
def test_time_upsample_matches_torch_bilinear_align_corners(B, T_in, T_out, C):
    from bsed_amd import ops
    g = torch.Generator().manual_seed(T_in)
    x = torch.randn(B, T_in, C, generator=g, requires_grad=True)
    ref = torch.nn.functional.interpolate(x.permute(0, 2, 1).unsqueeze(-1), size=(T_out, 1), mode="bilinear",
                                          align_corners=True).squeeze(-1).permute(0, 2, 1)
    dout = torch.randn(B, T_out, C, generator=g)
    ref.backward(dout)
    wide = torch.zeros(B, T_out, 2 * C, device="cuda")
    ops.upsample_time(x.detach().cuda(), T_out, out=wide, out_offset=C)
    # (source positions are computed in fp32 on both sides; a frame landing within 1e-6 of an integer position may pick
    # the neighbouring pair of inputs with weights (1-eps, eps), and one ulp of the position (1.5e-5 at frame 155) times
    # the local difference of the random test signal (up to ~6) is the error scale: 1e-4)
    assert float((wide[:, :, C:].cpu() - ref.detach()).abs().max()) < 1e-4 and float(wide[:, :, :C].abs().max()) == 0.0
    dwide = torch.zeros(B, T_out, 2 * C, device="cuda")
    dwide[:, :, C:] = dout.cuda()
    din = ops.upsample_time_bwd(dwide, T_in, C, in_offset=C)
    assert float((din.cpu() - x.grad).abs().max()) < 2e-4
