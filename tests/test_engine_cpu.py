"""Host logic of the PRODUCT package (no GPU): schedules and ramps of bsed_amd.engine against the vectors the
reference's own functions produced (tests/golden/schedules.json, written by oracle/gen_golden.py from
src/utilities/ramps.py and src/main_baseline.py:53-88), and the GRL warm-start coefficient of bsed_amd.disc."""
import json
import os

import numpy as np


def test_product_ramps_match_reference(golden_dir):
    from bsed_amd import engine
    ref = json.load(open(os.path.join(golden_dir, "schedules.json")))
    assert [engine.sigmoid_rampdown(e, 30) for e in range(0, 40, 3)] == ref["sigmoid_rampdown_30"]
    assert [engine.exp_rampup(e, 50) for e in range(0, 60, 5)] == ref["exp_rampup_50"]
    assert engine.exp_rampup(3, 0) == 1.0 and engine.sigmoid_rampdown(3, 0) == 1.0


class _TorchLikeOpt:
    def __init__(self):
        self.param_groups = [{"lr": 0.0}, {"lr": 0.0}]


class _FlatLikeOpt:
    lr = 0.0


def test_product_adjust_learning_rate_matches_reference(golden_dir):
    from bsed_amd import engine
    ref = json.load(open(os.path.join(golden_dir, "schedules.json")))
    for e, lr, lr_d, lr_c in ref["adjust_learning_rate"]:
        for mk in (_TorchLikeOpt, _FlatLikeOpt):
            o, od, oc = mk(), mk(), mk()
            got = engine.adjust_learning_rate(o, engine.sigmoid_rampdown(e, 30), optimizer_d=od, optimizer_crnn=oc,
                                              c_epoch=e)
            vals = [x.param_groups[-1]["lr"] if hasattr(x, "param_groups") else x.lr for x in (o, od, oc)]
            assert got == lr and vals == [lr, lr_d, lr_c], (e, vals)


def test_product_grl_coefficient_matches_reference(golden_dir):
    from bsed_amd import disc
    ref = json.load(open(os.path.join(golden_dir, "schedules.json")))
    got = [disc.grl_coeff(i) for i in range(5)]
    assert np.allclose(got, ref["grl_coeff_first5"], atol=1e-7)
