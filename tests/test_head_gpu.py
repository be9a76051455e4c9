"""Predictor-side helpers of the train loop that have no test of their own elsewhere.

  * ``ops.max_over_time`` is the train loop's ``target_weak = target.max(-2)[0]`` (reference src/main_baseline.py, train_mt:
    the clip-level targets of the synthetic batch are derived from the strong ones every step): a maximum is exact, so the
    bar is bit-equality with torch, on the bench shape, ragged / tiny shapes, negative values and a float64 input;
  * the head kernels at a frame count that is not a multiple of their 32-frame chunk and at one clip, against autograd
    on the same formulas (reference src/models/CRNN_GRL.py:441-460 and the BCE / MSE assembly of main_baseline.py:431-498).
"""
import pytest
import torch

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("B,T,C", [(256, 216, 20), (1, 1, 20), (3, 313, 20), (2, 7, 1), (5, 40, 128), (4, 9, 256)])
def test_max_over_time_is_exact(B, T, C):
    from bsed_amd import ops
    g = torch.Generator(device="cuda").manual_seed(B * 1000 + T)
    y = torch.randn((B, T, C), device="cuda", generator=g) - 3.0       # mostly negative: the identity must be -inf, not 0
    out = ops.max_over_time(y)
    assert out.shape == (B, C) and out.dtype == torch.float32
    assert torch.equal(out, y.max(-2)[0])
    hot = (torch.rand((B, T, C), device="cuda", generator=g) < 0.05).double()   # many-hot targets, float64 as numpy gives them
    assert torch.equal(ops.max_over_time(hot), hot.max(-2)[0].float())


@pytest.mark.parametrize("B,T", [(1, 77), (5, 216), (3, 31)])
def test_head_forward_backward_vs_autograd(B, T):
    from bsed_amd.models import Predictor, weights_init
    torch.manual_seed(3)
    pred = Predictor(nclass=20, attention=True, n_RNN_cell=128)
    weights_init(pred)
    with torch.no_grad():
        pred.flat.mul_(20.0)                                             # logits of order 1
    C, K = 20, 256
    g = torch.Generator(device="cuda").manual_seed(17)
    enc = torch.randn((B, T, K), device="cuda", generator=g)
    y = (torch.rand((B, T, C), device="cuda", generator=g) < 0.2).float()
    es = torch.rand((B, T, C), device="cuda", generator=g)
    ew = torch.rand((B, C), device="cuda", generator=g)
    pred.train()
    saved = pred.run_forward(enc)
    pred.flat_grad.zero_()
    from bsed_amd import ops
    yw = ops.max_over_time(y)
    dx, lp = pred.run_backward(enc, saved, y_strong=y, y_weak=yw, ema_strong=es, ema_weak=ew, w_cons_s=0.7, w_cons_w=0.3)
    # float64 autograd on the reference's formulas
    w = pred.flat[:2 * C * K].detach().double().view(2 * C, K).requires_grad_(True)
    b = pred.flat[2 * C * K:].detach().double().requires_grad_(True)
    x = enc.double().requires_grad_(True)
    lin = x @ w.t() + b
    strong = torch.sigmoid(lin[..., :C])
    sof = torch.softmax(lin[..., C:], dim=-1).clamp(1e-7, 1.0)
    weak = (strong * sof).sum(1) / sof.sum(1)
    bce = torch.nn.functional.binary_cross_entropy
    loss = bce(strong, y.double()) + bce(weak, yw.double()) \
        + 0.7 * torch.nn.functional.mse_loss(strong, es.double()) + 0.3 * torch.nn.functional.mse_loss(weak, ew.double())
    loss.backward()
    assert (saved[0].double() - strong).abs().max() < 2e-6
    assert (saved[2].double() - weak).abs().max() < 2e-6
    parts = lp.double().sum(0)
    mine = parts[0] / (B * T * C) + parts[1] / (B * C) + 0.7 * parts[2] / (B * T * C) + 0.3 * parts[3] / (B * C)
    assert abs(float(mine) - float(loss)) < 2e-6 * abs(float(loss))
    rel = lambda a, r: float((a.double() - r).norm() / r.norm())
    assert rel(dx, x.grad) < 2e-5
    assert rel(pred.flat_grad[:2 * C * K].view(2 * C, K), w.grad) < 2e-5
    assert rel(pred.flat_grad[2 * C * K:], b.grad) < 2e-5
