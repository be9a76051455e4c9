#!/usr/bin/env python3
"""How far the "bf16" throughput mode (bf16 CNN activations, one bf16 MFMA per product) is from the fp32 oracle:
forward outputs, loss and every gradient tensor, at the reference configuration (B = 2, 1255 frames) and at a batch of
B clips of 865 frames; the split-fp32 mode beside it.   python tools/bf16_margin.py [B]"""
import os
import sys

import numpy as np
import torch

R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, R)
sys.path.insert(0, os.path.join(R, "tests"))
from oracle import crnn_oracle as co  # noqa: E402
from oracle import seeded  # noqa: E402
from test_crnn_gpu import _mine, _oracle  # noqa: E402


def run(B, T, seed, modes=("bf16x3", "bf16")):
    x = torch.from_numpy(seeded.db_like_input(seed + 10, B, T))
    y = torch.from_numpy(seeded.strong_targets(seed + 11, B, T // 4))
    ocrnn, opred = _oracle(0.0, seed)
    ocrnn.train(); opred.train()
    loss_ref, out_ref = co.train_losses(ocrnn, opred, x, y)
    loss_ref.backward()
    for mode in modes:
        crnn, pred = _mine(0.0, ocrnn, opred, mode)
        crnn.train(); pred.train()
        enc, ctx = crnn.run_forward(x.cuda(), save=True)
        saved = pred.run_forward(enc)
        strong, sof, weak, den = saved
        yw = y.max(-2)[0]
        crnn.zero_grad(); pred.zero_grad()
        dx, loss_part = pred.run_backward(enc, saved, y_strong=y.cuda(), y_weak=yw.cuda())
        lp = loss_part.sum(0).cpu().double()
        loss = float(lp[0] / (B * (T // 4) * 20) + lp[1] / (B * 20))
        crnn.run_backward(ctx, dx)
        e_enc = float((enc.cpu() - out_ref["enc_syn"].detach()).abs().max())
        e_enc_rel = float((enc.cpu() - out_ref["enc_syn"].detach()).norm() / out_ref["enc_syn"].detach().norm())
        e_s = float((strong.cpu() - out_ref["strong_syn"].detach()).abs().max())
        e_w = float((weak.cpu() - out_ref["weak_syn"].detach()).abs().max())
        worst = []
        for mod, omod in ((crnn, ocrnn), (pred, opred)):
            for k, p in omod.named_parameters():
                if ".conv" in k and k.endswith(".bias"):
                    continue
                got = mod.P(k.replace("cnn.cnn.", "cnn.", 1)).grad.detach().cpu().double()
                ref = p.grad.double()
                worst.append((float((got - ref).norm() / (ref.norm() + 1e-30)), k))
        worst.sort(reverse=True)
        print(f"B={B} T={T} {mode:7s}: enc max abs {e_enc:.3g} (rel L2 {e_enc_rel:.3g}), strong {e_s:.3g}, weak {e_w:.3g}, "
              f"loss rel {abs(loss - float(loss_ref)) / abs(float(loss_ref)):.3g}; grad rel-L2: worst "
              + ", ".join(f"{k} {v:.3g}" for v, k in worst[:4]) + f"; median {np.median([v for v, _ in worst]):.3g}", flush=True)


run(2, 1255, 3)
run(int(sys.argv[1]) if len(sys.argv) > 1 else 8, 865, 5)
