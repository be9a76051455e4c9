#!/usr/bin/env python3
"""python tools/step_ab.py out.npz [B T steps] -- a few CRNN train steps (dropout on, Adam) from seeded inputs; weights and
losses after every step go to out.npz.  Run once per build / environment switch and compare with tools/npz_equal.py:
changes that only move WHEN or WHERE the same arithmetic runs (launch fusion, streams) must give identical bits."""
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from oracle import crnn_oracle as co, seeded  # noqa: E402  (test infrastructure: seeded inputs only)
from bsed_amd.engine import FlatAdam, SEDTrainer  # noqa: E402
from bsed_amd.models import CRNN, Predictor, weights_init  # noqa: E402

out = sys.argv[1]
B, T, steps = (int(v) for v in sys.argv[2:5]) if len(sys.argv) > 4 else (24, 865, 3)
kw = dict(co.CRNN_KWARGS); kw["dropout"] = 0.5
torch.manual_seed(3)
crnn, pred = CRNN(**kw), Predictor(**co.PREDICTOR_KWARGS)
weights_init(crnn); weights_init(pred)
x = torch.from_numpy(seeded.db_like_input(12, B, T)).cuda()
y = torch.from_numpy(seeded.strong_targets(13, B, T // 4)).cuda()
tr = SEDTrainer(crnn, pred, optimizer=FlatAdam([crnn, pred], lr=1e-3), seed=7)
res = {}
for i in range(steps):
    o = tr.train_step(x, y)
    res[f"loss{i}"] = np.float64(SEDTrainer.loss_value(o))
    res[f"crnn{i}"] = crnn.flat.cpu().numpy(); res[f"pred{i}"] = pred.flat.cpu().numpy()
    res[f"bufs{i}"] = crnn.flat_buf.cpu().numpy()
np.savez(out, **res)
print("wrote", out)
