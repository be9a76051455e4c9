#!/usr/bin/env python3
"""time the split-fp32 GLU forward / backward of one block shape at B = 256 (grid-size knobs: BSED_GLU_FWD3_G32/64/128,
BSED_GLU_BWD3_G32/64 are read once per process)   python tools/glu_time.py C H W ph pw"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from bsed_amd import ops
C, H, W, ph, pw = (int(v) for v in sys.argv[1:6])
B = 256
g = torch.Generator(device="cuda").manual_seed(3)
y = torch.randn(B, H, W, C, device="cuda", generator=g)
sc = torch.rand(C, device="cuda", generator=g) + 0.5
sh = torch.randn(C, device="cuda", generator=g) * 0.1
w = torch.randn(C, C, device="cuda", generator=g) * 0.1
b = torch.randn(C, device="cuda", generator=g) * 0.1
dp = torch.randn(B, H // ph, W // pw, C, device="cuda", generator=g) * 1e-3
def timed(fn, reps=10):
    fn(); torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(reps): fn()
    e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e) / reps * 1e3
f = timed(lambda: ops.glu_fwd3(y, sc, sh, w, b, B, H, W, C, (ph, pw), 0.5, 101, 7))
bw = timed(lambda: (ops.glu_bwd3n if C == 128 else ops.glu_bwd3)(y, sc, sh, w, b, dp, B, H, W, C, (ph, pw), 0.5, 101, 7))
print(f"C={C} {H}x{W}: fwd {f:.1f} us  bwd {bw:.1f} us   env " + " ".join(f"{k}={v}" for k, v in os.environ.items() if k.startswith("BSED_GLU")))
