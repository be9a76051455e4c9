#!/usr/bin/env python3
"""python tools/head_ab.py out.npz [B T] -- Predictor forward + backward (strong / weak BCE + consistency terms) on seeded
inputs; run once per library build (BSED_LIB_PATH) and compare the files with tools/npz_equal.py: a kernel edit that only
changes HOW operands are fetched must give identical bits.
(Round 2, fourth session: float4 LDS reads over k in the head kernels -- pitch 260, 6 ds_read_b128 per 20 FMAs instead of 24
ds_read_b32 -- gave identical bits and were SLOWER: forward 102 -> 113 us, backward 135 -> 202 us; not kept.)"""
import sys

import numpy as np
import torch

sys.path.insert(0, __import__("os").path.dirname(__import__("os").path.dirname(__import__("os").path.abspath(__file__))))
from bsed_amd.models import Predictor, weights_init  # noqa: E402

out = sys.argv[1]
B, T = (int(sys.argv[2]), int(sys.argv[3])) if len(sys.argv) > 3 else (256, 216)
torch.manual_seed(7)
pred = Predictor(nclass=20, attention=True, n_RNN_cell=128)
weights_init(pred)
with torch.no_grad():
    pred.flat.mul_(20.0)     # logits of order 1
g = torch.Generator(device="cuda").manual_seed(11)
enc = torch.randn((B, T, 256), device="cuda", generator=g)
y = (torch.rand((B, T, 20), device="cuda", generator=g) < 0.2).float()
yw = y.max(1)[0].contiguous()
es = torch.rand((B, T, 20), device="cuda", generator=g)
ew = torch.rand((B, 20), device="cuda", generator=g)
res = {}
pred.train()
saved = pred.run_forward(enc)
for i, t in enumerate(saved):
    if t is not None:
        res[f"fwd{i}"] = t.cpu().numpy()
pred.flat_grad.zero_()
dx, lp = pred.run_backward(enc, saved, y_strong=y, y_weak=yw, ema_strong=es, ema_weak=ew, w_cons_s=1.0, w_cons_w=1.0)
torch.cuda.synchronize()
res["dx"] = dx.cpu().numpy(); res["loss_parts"] = lp.cpu().numpy(); res["grad"] = pred.flat_grad.cpu().numpy()
np.savez(out, **res)
print("wrote", out, {k: v.shape for k, v in res.items()})
