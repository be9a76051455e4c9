#!/usr/bin/env python3
"""Phase stamps of wgrad3p_kernel's fourth tile period (diagnostic build: tools/build_variant.sh w3pstamp igemm.hip
"-DW3P_STAMP", run with BSED_LIB_PATH=.../libbsed_w3pstamp.so): shader cycles, median over the workgroups of a launch.
    python tools/wgrad_stamp.py [W] [CIN] [N]"""
import ctypes
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from bsed_amd import ops, _lib  # noqa: E402

W = int(sys.argv[1]) if len(sys.argv) > 1 else 8
CIN = int(sys.argv[2]) if len(sys.argv) > 2 else 128
N = int(sys.argv[3]) if len(sys.argv) > 3 else 128
B, H = 256, 216
x = torch.randn(B, H, W, CIN, device="cuda")
g = torch.randn(B, H, W, N, device="cuda")
y = torch.randn(B, H, W, N, device="cuda")
coef = torch.randn(3, N, device="cuda")
mean = torch.randn(N, device="cuda")
for bn in (False, True):
    for _ in range(3):
        if bn:
            ops.wgrad(x, g, B, H, W, CIN, N, taps=ops.TAPS3x3, bn_y=y, bn_coef=coef, bn_mean=mean, dy_out=torch.empty_like(g))
        else:
            ops.wgrad(x, g, B, H, W, CIN, N, taps=ops.TAPS3x3)
    torch.cuda.synchronize()
    buf = (ctypes.c_ulonglong * (1024 * 8))()
    assert _lib.lib().bsed_w3p_stamps(buf) == 0
    t = np.array(buf, dtype=np.int64).reshape(1024, 8)
    t = t[t[:, 0] > 0][:256]
    d = lambda a, b: np.median(t[:, b] - t[:, a])
    print(f"W={W} {CIN}->{N} bn={bn}: {len(t)} workgroups; producer: wait for loads {d(0, 1):.0f}, convert+refill {d(1, 2):.0f}, "
          f"barrier {d(2, 3):.0f}, period {d(0, 3):.0f} | consumer: mfma loop {d(4, 5):.0f}, barrier {d(5, 6):.0f}, period {d(4, 6):.0f} "
          f"| producer start - consumer start {np.median(t[:, 0] - t[:, 4]):.0f}")
