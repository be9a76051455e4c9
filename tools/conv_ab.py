#!/usr/bin/env python3
"""A/B of the split-fp32 3x3 convolution kernels on the network's layer shapes (B = 256, 22.05 kHz):
    slab kernel of rounds 1-2 (csrc/igemm3.hip)  vs  N-split kernel (csrc/igemm3n.hip) at 2 / 3 waves per SIMD.
Outputs are compared bit for bit, the BatchNorm partial sums in float64; times are medians over interleaved rounds in
ONE process (guide rule 24).   python tools/conv_ab.py [rounds] [B]"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from bsed_amd import ops  # noqa: E402

rounds = int(sys.argv[1]) if len(sys.argv) > 1 else 7
B = int(sys.argv[2]) if len(sys.argv) > 2 else 256
# knob values of ops.set_igemm3n_wpe: 2 / 3 / 4 (+ 8: raised priority outside the loop), 0 = the library's default
VARIANTS = [int(v) for v in sys.argv[3].split(",")] if len(sys.argv) > 3 else [0, 2, 3]
dev = "cuda"
T = 216
# (name, H, W, CIN, N, stats, taps)
fl = [(-a, -b) for a, b in ops.TAPS3x3]
SHAPES = [
    ("fwd2 32->64", T, 32, 32, 64, True, ops.TAPS3x3),
    ("fwd3 64->128", T, 16, 64, 128, True, ops.TAPS3x3),
    ("fwd4 128->128", T, 8, 128, 128, True, ops.TAPS3x3),
    ("fwd5 128->128", T, 4, 128, 128, True, ops.TAPS3x3),
    ("fwd6 128->128", T, 2, 128, 128, True, ops.TAPS3x3),
    ("dgr6 128->128", T, 2, 128, 128, False, fl),
    ("dgr5 128->128", T, 4, 128, 128, False, fl),
    ("dgr4 128->128", T, 8, 128, 128, False, fl),
    ("dgr3 128->64", T, 16, 128, 64, False, fl),
    ("dgr2 64->32", T, 32, 64, 32, False, fl),
    ("gru xp 128->768", B * T, 1, 128, 768, False, ((0, 0),)),
    ("gru xp 256->768", B * T, 1, 256, 768, False, ((0, 0),)),
    ("gru d 768->256", B * T, 1, 768, 256, False, ((0, 0),)),
]


def timed(fn):
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    fn()
    e.record()
    torch.cuda.synchronize()
    return s.elapsed_time(e)


tot = {}
for name, H, W, CIN, N, st, taps in SHAPES:
    nb = 1 if name.startswith("gru") else B
    x = torch.randn(nb, H, W, CIN, device=dev)
    w = torch.randn(len(taps), CIN, N, device=dev) * 0.05     # [tap][k][n]
    bias = torch.randn(N, device=dev)
    epi = ops.EPI_STATS if st else ops.EPI_PLAIN
    os.environ["BSED_IGEMM3N"] = "0"
    w_old = ops.pack_weight3(w, len(taps), CIN, N, CIN * N, N, 1)
    os.environ["BSED_IGEMM3N"] = "1"
    w_new = ops.pack_weight3(w, len(taps), CIN, N, CIN * N, N, 1)
    variants = {"slab": (w_old, None)}
    for v in VARIANTS:
        variants[f"n/{v}"] = (w_new, v)

    def run(v):
        wt, wpe = variants[v]
        if wpe is not None:
            ops.set_igemm3n_wpe(wpe)
        return ops.igemm3(x, wt, N, nb, H, W, CIN, taps, bias=bias, epilogue=epi)

    ref, ref_st = run("slab")
    line = f"{name:18s}"
    for v in variants:
        out, stt = run(v)
        torch.cuda.synchronize()
        same = torch.equal(out, ref)
        sterr = 0.0
        if st:
            a, b = stt.double().sum(0), ref_st.double().sum(0)
            sterr = float(((a - b).abs() / (b.abs() + 1e-3)).max())
        if not same or sterr > 1e-5:
            line += f" !! {v}: out bits {'same' if same else 'DIFFER max %.3g' % float((out - ref).abs().max())} stats rel {sterr:.2g}"
    ts = {v: [] for v in variants}
    for _ in range(rounds):
        for v in variants:
            ts[v].append(timed(lambda: run(v)))
    flops = 2.0 * nb * H * W * len(taps) * CIN * N
    for v in variants:
        m = sorted(ts[v])[len(ts[v]) // 2]
        tot[v] = tot.get(v, 0.0) + m
        line += f"  {v} {m * 1e3:7.1f} us ({flops / m / 1e9:5.0f} TF)"
    print(line, flush=True)
print("sum of medians:", "  ".join(f"{v} {t:.3f} ms" for v, t in tot.items()))
