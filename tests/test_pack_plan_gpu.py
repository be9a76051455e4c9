"""Packed weight copies of a train step made in ONE launch at its start (ops.PackPlan, bsed_pack_weights_batch).

The reference has no counterpart (its convolutions read the PyTorch weight tensors, src/models/CNN.py:46-47); the
property checked is that the plan changes WHEN the bf16 hi/lo copies are made, never their bits:

  * the batched kernel writes the same bytes as the single-job entries (both layouts, several shapes, strided sources);
  * three train steps with the plan and three without give the same losses and bit-identical weights (plain, mean
    teacher with the teacher on its own stream, adversarial);
  * the plan holds exactly what a step used (entries of a path that is no longer taken are dropped).
"""
import numpy as np
import pytest
import torch

from oracle import crnn_oracle as co
from oracle import seeded

pytestmark = pytest.mark.gpu


def test_batched_pack_writes_the_same_bytes():
    from bsed_amd import ops
    torch.manual_seed(5)
    specs = [  # (kind, weight shape (N, K, taps...), ntaps, K, N, strides)
        ("w3", (64, 32, 3, 3), 9, 32, 64, (1, 9, 32 * 9)),        # conv forward: src[n][k][tap]
        ("w3", (128, 64, 3, 3), 9, 128, 64, (1, 64 * 9, 9)),      # data gradient: roles of n and k swapped
        ("w3", (768, 128), 1, 128, 768, (0, 1, 128)),             # GRU input projection
        ("w3", (768, 256), 1, 768, 256, (0, 256, 1)),             # its transpose
        ("w3", (20, 256), 1, 256, 20, (0, 1, 256)),               # N below a 32 multiple: zero padding
        ("w3s", (32, 16, 3, 3), 9, 16, 32, (1, 9, 16 * 9)),
        ("w3s", (32, 16, 3, 3), 9, 32, 16, (1, 16 * 9, 9)),
    ]
    weights = [torch.randn(s[1], device="cuda") for s in specs]

    def pack_all():
        out = []
        for (kind, _, ntaps, K, N, (st, sk, sn)), w in zip(specs, weights):
            if kind == "w3":
                out.append(ops.pack_weight3(w, ntaps, K, N, st, sk, sn))
            else:
                out.append(ops.pack_weight3s(w, ntaps, N, st, sk, sn, K=K))
        return out

    single = pack_all()
    plan = ops.PackPlan()
    with ops.pack_cache(plan):
        first = pack_all()                      # recorded, packed one by one
    assert len(plan.entries) == len(specs)
    with ops.pack_cache(plan):
        assert len(ops._pack_memo) == len(specs)   # all made on entry, in one launch
        planned = pack_all()
        base = planned[0].untyped_storage().data_ptr()
        assert all(p.untyped_storage().data_ptr() == base for p in planned)   # ... into one buffer
    for a, b, c in zip(single, first, planned):
        assert a.shape == b.shape == c.shape
        assert torch.equal(a, b) and torch.equal(a, c)
    # a block that uses only part of the plan drops the rest
    with ops.pack_cache(plan):
        ops.pack_weight3(weights[0], 9, 32, 64, 1, 9, 32 * 9)
    assert len(plan.entries) == 1


def _models(dropout, seed=3):
    from bsed_amd.models import CRNN, Predictor, weights_init
    kw = dict(co.CRNN_KWARGS)
    kw["dropout"] = dropout
    torch.manual_seed(seed)
    crnn, pred = CRNN(**kw), Predictor(**co.PREDICTOR_KWARGS)
    weights_init(crnn); weights_init(pred)
    return crnn, pred


@pytest.mark.parametrize("mode", ["crnn", "mt", "ada"])
def test_planned_steps_are_bit_identical(mode):
    from bsed_amd.engine import FlatAdam, FlatSGD, SEDTrainer
    from bsed_amd.models import CRNN, Predictor
    nb, T = 6, 256
    xs = torch.from_numpy(seeded.db_like_input(41, nb, T)).cuda()
    xr = torch.from_numpy(seeded.db_like_input(42, nb, T)).cuda()
    ys = torch.from_numpy(seeded.strong_targets(43, nb, T // 4)).cuda()
    yw = ys.max(1)[0].contiguous()
    runs = []
    for planned in (False, True):
        crnn, pred = _models(0.5)
        extra = {}
        if mode == "mt":
            kw = dict(co.CRNN_KWARGS); kw["dropout"] = 0.5
            ema_c, ema_p = CRNN(**kw), Predictor(**co.PREDICTOR_KWARGS)
            ema_c.load_state_dict(crnn.state_dict()); ema_p.load_state_dict(pred.state_dict())
            extra = dict(ema_crnn=ema_c, ema_predictor=ema_p)
        elif mode == "ada":
            from bsed_amd.disc import Clip_Discriminator, ConditionalDomainAdversarialLoss
            torch.manual_seed(9)
            disc = Clip_Discriminator()
            extra = dict(domain_loss=ConditionalDomainAdversarialLoss(disc),
                         optimizer_d=FlatSGD([disc], lr=1e-4, momentum=0.9, weight_decay=1e-4, nesterov=True))
        tr = SEDTrainer(crnn, pred, optimizer=FlatAdam([crnn, pred], lr=1e-3), seed=11, **extra)
        if not planned:
            tr._pack_plans = None
        trace = []
        for i in range(3):
            if mode == "crnn":
                out = tr.train_step(xs, ys)
            else:
                out = tr.train_step(xs, ys, xr, yw if mode == "mt" else None)
            trace.append((SEDTrainer.loss_value(out), crnn.flat.clone(), pred.flat.clone()))
        if planned:
            step = tr._pack_plans["step"]
            assert len(step.entries) >= 12          # 6 + 5 convolution layouts, 2 + 2 GRU projections (fewer launches
            assert step.used == set(step.entries)   # than that means the plan is not in use)
            if mode == "mt":
                assert len(tr._pack_plans["teacher"].entries) >= 6
        runs.append(trace)
    for (la, ca, pa), (lb, cb, pb) in zip(*runs):
        assert la == lb
        assert torch.equal(ca, cb) and torch.equal(pa, pb)
