#!/usr/bin/env python3
"""The fused first block (csrc/block0.hip) in isolation at the bench shape: event-timed launches, and a target for
`rocprofv3 --pmc ...` passes.   python tools/block0_bench.py [reps] [B] [which: all|stats|fwd|bwd]"""
import os
import sys
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from bsed_amd import ops  # noqa: E402

reps = int(sys.argv[1]) if len(sys.argv) > 1 else 5
B = int(sys.argv[2]) if len(sys.argv) > 2 else 256
which = sys.argv[3] if len(sys.argv) > 3 else "all"
H, W = 865, 128
dev = "cuda"
g = torch.Generator(device=dev).manual_seed(1)
x = (torch.rand(B, H, W, device=dev, generator=g) * 60 - 70)
cw = torch.randn(16, 1, 3, 3, device=dev, generator=g) * 0.3
cb = torch.randn(16, device=dev, generator=g) * 0.1
wg = torch.randn(16, 16, device=dev, generator=g) * 0.2
bg = torch.randn(16, device=dev, generator=g) * 0.1
scale = torch.rand(16, device=dev, generator=g) * 0.05 + 0.02
shift = torch.randn(16, device=dev, generator=g) * 0.1
dpool = torch.randn(B, H // 2, W // 2, 16, device=dev, generator=g) * 1e-3


def timed(name, fn):
    fn(); torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(reps):
        fn()
    e.record(); torch.cuda.synchronize()
    print(f"{name:10s} {s.elapsed_time(e) / reps:8.3f} ms")


if which in ("all", "stats"):
    timed("stats", lambda: ops.block0_stats(x, cw, cb, B, H, W))
if which in ("all", "fwd"):
    timed("fwd", lambda: ops.block0_fwd(x, cw, cb, scale, shift, wg, bg, B, H, W, (2, 2), 0.5, 100, 7))
if which in ("all", "bwd"):
    timed("bwd", lambda: ops.block0_bwd(x, cw, cb, scale, shift, wg, bg, dpool, B, H, W, (2, 2), 0.5, 100, 7))
