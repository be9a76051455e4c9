"""BASELINE configs[1] -- CNN-only tagging forward: the product ``CRNN_pred`` (CNN stack -> sigmoid features ->
class-softmax attention pooling, csrc/tag.hip) against vectors produced by RUNNING the reference's own ``CRNN_pred``
(src/models/CRNN_GRL.py:206-290; tests/golden/cnn_pred.npz written by oracle/gen_golden.py), in both contraction
modes, plus the batch-64 shape of the BASELINE configuration against the CPU oracle on two of its clips."""
import os

import numpy as np
import pytest
import torch

from oracle import crnn_oracle as co
from oracle import seeded

pytestmark = pytest.mark.gpu


def _kw(dropout):
    kw = dict(co.CRNN_KWARGS)
    kw.update(nclass=128, n_RNN_cell=64, dropout=dropout)
    return kw


@pytest.mark.parametrize("mode", ["bf16x3", "fp32"])
@pytest.mark.parametrize("tag", ["small", "R"])
def test_cnn_pred_forward_matches_reference(golden_dir, tag, mode):
    from bsed_amd.models import CRNN_pred
    g = np.load(os.path.join(golden_dir, "cnn_pred.npz"))
    B, T, seed = (int(v) for v in g[f"{tag}_meta"])
    x = torch.from_numpy(seeded.db_like_input(seed + 10, B, T)).cuda()
    m = CRNN_pred(**_kw(0.5))
    assert sorted(m.state_dict().keys()) == sorted(str(n) for n in g["state_names"])
    shapes = {str(n): str(s) for n, s in zip(g["state_names"], g["state_shapes"])}
    assert all(str(tuple(v.shape)) == shapes[k] for k, v in m.state_dict().items())
    vals = seeded.seeded_state({str(n): eval(s) for n, s in shapes.items()}, seed)
    assert seeded.checksum(vals) == float(g[f"{tag}_weight_checksum"][0])
    m.load_state_dict({k: torch.from_numpy(v) for k, v in vals.items()})
    m.conv_mode = mode
    m.eval()
    with torch.no_grad():
        strong, weak = m(x)
    assert float((strong.cpu() - torch.from_numpy(g[f"{tag}_eval_strong"])).abs().max()) < 2e-5
    assert float((weak.cpu() - torch.from_numpy(g[f"{tag}_eval_weak"])).abs().max()) < 2e-5
    # inference=True: strong masked by weak > 0.5 (reference :282-287)
    with torch.no_grad():
        s2, w2 = m(x, inference=True)
    want = g[f"{tag}_eval_strong"] * (g[f"{tag}_eval_weak"] > 0.5)[:, None, :]
    edge = np.abs(g[f"{tag}_eval_weak"] - 0.5) < 1e-4          # classes whose weak sits on the threshold
    got = s2.cpu().numpy()
    assert np.abs(got - want)[:, :, ~edge.any(0)].max() < 2e-5
    # train mode, dropout 0: BatchNorm batch statistics + running-stat update
    m = CRNN_pred(**_kw(0.0))
    m.load_state_dict({k: torch.from_numpy(v) for k, v in vals.items()})
    m.conv_mode = mode
    m.train()
    with torch.no_grad():
        strong, weak = m(x)
    assert float((strong.cpu() - torch.from_numpy(g[f"{tag}_train_strong"])).abs().max()) < 2e-5
    assert float((weak.cpu() - torch.from_numpy(g[f"{tag}_train_weak"])).abs().max()) < 2e-5
    sd = m.state_dict()
    np.testing.assert_allclose(sd["cnn.batchnorm6.running_var"].cpu().numpy(), g[f"{tag}_after_rv6"], rtol=2e-4)
    np.testing.assert_allclose(sd["cnn.batchnorm6.running_mean"].cpu().numpy(), g[f"{tag}_after_rm6"], rtol=2e-4,
                               atol=1e-5)


def test_cnn_pred_batch64_bench_shape_vs_oracle():
    """bench.py --mode cnn: 64 clips x 865 frames, eval forward; two clips against the oracle, all against themselves
    run alone (per-clip independence in eval mode)"""
    from bsed_amd.models import CRNN_pred
    B, T, seed = 64, 865, 71
    x = torch.from_numpy(seeded.db_like_input(seed, B, T)).cuda()
    om = co.CRNN_pred(**_kw(0.5))
    seeded.load_seeded(om, seed + 1)
    om.eval()
    m = CRNN_pred(**_kw(0.5))
    m.load_state_dict(om.state_dict())
    m.eval()
    with torch.no_grad():
        strong, weak = m(x)
        assert strong.shape == (B, T // 4, 128) and weak.shape == (B, 128)
        so, wo = om(x[[3, 40]].cpu())
        for i, b in enumerate((3, 40)):
            assert float((strong[b].cpu() - so[i]).abs().max()) < 2e-5
            assert float((weak[b].cpu() - wo[i]).abs().max()) < 2e-5
        s1, w1 = m(x[17:18])
        assert float((s1[0] - strong[17]).abs().max()) < 1e-5 and float((w1[0] - weak[17]).abs().max()) < 1e-5


def test_cnn_pred_is_forward_only_and_checks_config():
    from bsed_amd.models import CRNN_pred
    with pytest.raises(NotImplementedError):
        kw = _kw(0.5); kw["nclass"] = 20
        CRNN_pred(**kw)
    m = CRNN_pred(**_kw(0.5))
    with pytest.raises(NotImplementedError):
        m.run_backward(None, None)
