#!/usr/bin/env python3
"""A/B harness for the weight-gradient kernel: runs a fixed set of layer shapes and writes the results to an .npz, so
two builds of the library (BSED_LIB_PATH=...) can be compared bit for bit and timed against each other.
    python tools/wgrad_ab.py out.npz"""
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from bsed_amd import ops  # noqa: E402

# (B, H, W, CIN, N, taps) -- the seven conv layers' gradients at a reduced batch + the odd shapes of the unit tests
SHAPES = [(64, 432, 64, 16, 32, 9), (64, 216, 32, 32, 64, 9), (64, 216, 16, 64, 128, 9), (64, 216, 8, 128, 128, 9),
          (64, 216, 4, 128, 128, 9), (64, 216, 2, 128, 128, 9), (64, 216, 1, 128, 128, 3), (1, 64 * 216 * 8, 1, 128, 128, 1),
          (3, 50, 8, 64, 64, 9), (2, 37, 16, 32, 32, 9), (1, 1000, 1, 256, 128, 1), (2, 100, 4, 64, 128, 9)]
out = {}
g = torch.Generator(device="cuda").manual_seed(7)
for i, (B, H, W, C, N, nt) in enumerate(SHAPES):
    x = torch.randn(B, H, W, C, device="cuda", generator=g)
    dy = torch.randn(B, H, W, N, device="cuda", generator=g)
    if nt == 9:
        taps = ops.TAPS3x3
    elif nt == 3:
        taps = [(-1, 0), (0, 0), (1, 0)]
    else:
        taps = None
    run = (lambda: ops.wgrad(x, dy, B, H, W, C, N, taps=taps)) if taps is not None else (lambda: ops.wgrad(x, dy, B, H, W, C, N))
    r = run()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(5):
        run()
    e.record()
    torch.cuda.synchronize()
    part = r[0]
    out[f"s{i}"] = part.sum(0).cpu().numpy()
    out[f"c{i}"] = part.view(torch.int32).long().sum().cpu().numpy()          # bit-level checksum of every slab
    print(f"{(B, H, W, C, N, nt)}: {s.elapsed_time(e) / 5:.3f} ms  G={r[1]}")
np.savez(sys.argv[1], **out)
