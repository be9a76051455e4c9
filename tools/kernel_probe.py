#!/usr/bin/env python3
"""Run ONE hot kernel shape in isolation (for rocprofv3 --pmc passes and A/B timing).
    python tools/kernel_probe.py wgrad9|conv|glubwd|wgrad1 [reps]"""
import sys
import os
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from bsed_amd import ops  # noqa: E402

which = sys.argv[1] if len(sys.argv) > 1 else "wgrad9"
if os.environ.get("BSED_IGEMM3_RB"):
    ops.IGEMM3_RB["rb"] = int(os.environ["BSED_IGEMM3_RB"])
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 5
B, H, W, C = 256, 216, 8, 128
dev = "cuda"
x = torch.randn(B, H, W, C, device=dev)
dy = torch.randn(B, H, W, C, device=dev)
w = torch.randn(C, C, 3, 3, device=dev) * 0.05
x16 = torch.randn(B, 432, 64, 16, device=dev) if which == "wgrad16" else None
dy32 = torch.randn(B, 432, 64, 32, device=dev) if which == "wgrad16" else None


def run():
    if which == "wgrad9":
        return ops.wgrad(x, dy, B, H, W, C, C, taps=ops.TAPS3x3)
    if which == "conv":
        wpk = ops.pack_weight(w, 9, C, C, 1, 9, C * 9)
        return ops.igemm(x, wpk, C, B, H, W, C, taps=ops.TAPS3x3, epilogue=ops.EPI_STATS)
    if which == "conv3":  # split-fp32 3x3 conv forward + BN partial sums (igemm3), the layer-4 shape
        w3 = ops.pack_weight3(w, 9, C, C, 1, 9, C * 9)
        return ops.igemm3(x, w3, C, B, H, W, C, ops.TAPS3x3, epilogue=ops.EPI_STATS)
    if which == "wgrad16":  # conv1's weight gradient: 16 -> 32 channels on the 432 x 64 map
        return ops.wgrad(x16, dy32, B, 432, 64, 16, 32, taps=ops.TAPS3x3)
    if which == "wgrad1":
        return ops.wgrad(x, dy, 1, B * H * W, 1, C, C)
    raise SystemExit("unknown kernel")


run()
torch.cuda.synchronize()
s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
s.record()
for _ in range(reps):
    run()
e.record()
torch.cuda.synchronize()
ms = s.elapsed_time(e) / reps
flops = 2.0 * B * H * W * C * C * (9 if which in ("wgrad9", "conv", "conv3") else 1)
if which == "wgrad16":
    flops = 2.0 * B * 432 * 64 * 9 * 16 * 32
print(f"{which}: {ms:.3f} ms  {flops / ms / 1e9:.1f} TFLOP/s")
