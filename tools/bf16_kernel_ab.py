#!/usr/bin/env python3
"""fp32 (split-fp32) vs bf16-activation instances of the CNN kernels on one layer shape, isolated (B = 256, 216 x W,
C channels): conv forward, GLU forward / backward, conv weight gradient with the BatchNorm map, conv data gradient.
    python tools/bf16_kernel_ab.py [W] [C]"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from bsed_amd import ops  # noqa: E402

W = int(sys.argv[1]) if len(sys.argv) > 1 else 8
C = int(sys.argv[2]) if len(sys.argv) > 2 else 128
B, H = 256, 216
g = torch.Generator(device="cuda").manual_seed(1)


def timed(fn, reps=8):
    fn(); torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(reps):
        fn()
    e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e) / reps * 1e3


for dt in (torch.float32, torch.bfloat16):
    x = torch.randn(B, H, W, C, device="cuda", generator=g).to(dt)
    y = torch.randn(B, H, W, C, device="cuda", generator=g).to(dt)
    gg = (torch.randn(B, H, W, C, device="cuda", generator=g) * 1e-3).to(dt)
    dp = (torch.randn(B, H, W // 2, C, device="cuda", generator=g) * 1e-3).to(dt)
    w = torch.randn(C, C, 3, 3, device="cuda", generator=g) * 0.05
    wl = torch.randn(C, C, device="cuda", generator=g) * 0.1
    bias = torch.randn(C, device="cuda", generator=g) * 0.1
    sc = torch.rand(C, device="cuda", generator=g) + 0.5
    sh = torch.randn(C, device="cuda", generator=g) * 0.1
    coef = torch.randn(3, C, device="cuda", generator=g)
    mean = torch.randn(C, device="cuda", generator=g)
    w3 = ops.pack_weight3(w, 9, C, C, 1, 9, C * 9)
    dy = torch.empty_like(gg)
    res = {
        "conv fwd": timed(lambda: ops.igemm3(x, w3, C, B, H, W, C, ops.TAPS3x3, bias=bias, epilogue=ops.EPI_STATS)),
        "glu fwd": timed(lambda: ops.glu_fwd3(y, sc, sh, wl, bias, B, H, W, C, (1, 2), 0.5, 101, 7)),
        "glu bwd": timed(lambda: (ops.glu_bwd3n if C == 128 else ops.glu_bwd3)(y, sc, sh, wl, bias, dp, B, H, W, C, (1, 2), 0.5, 101, 7)),
        "wgrad9+bn": timed(lambda: ops.wgrad(x, gg, B, H, W, C, C, taps=ops.TAPS3x3, bn_y=y, bn_coef=coef, bn_mean=mean, dy_out=dy)),
        "wgrad9": timed(lambda: ops.wgrad(x, gg, B, H, W, C, C, taps=ops.TAPS3x3)),
        "wgrad1": timed(lambda: ops.wgrad(y, gg, B, H, W, C, C, a_scale=sc, a_shift=sh)),
    }
    print(f"W={W} C={C} {str(dt):15s}: " + "  ".join(f"{k} {v:7.1f} us" for k, v in res.items()), flush=True)
