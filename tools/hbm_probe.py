#!/usr/bin/env python3
"""What this box's HBM delivers to plain streaming kernels (context for the roofline fractions in DESIGN.md):
read-only (sum), read+write (copy) and write-only (fill) over a 2 GiB fp32 buffer."""
import torch

n = 512 * 1024 * 1024
x = torch.empty(n, device="cuda", dtype=torch.float32).normal_()
y = torch.empty_like(x)


def timed(fn, nbytes, name, reps=10):
    fn(); torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(reps):
        fn()
    e.record(); torch.cuda.synchronize()
    ms = s.elapsed_time(e) / reps
    print(f"{name}: {ms:.3f} ms  {nbytes / ms / 1e9:.2f} TB/s")


timed(lambda: x.sum(), 4 * n, "read  (sum)")
timed(lambda: y.copy_(x), 8 * n, "copy  (read+write)")
timed(lambda: y.fill_(1.0), 4 * n, "write (fill)")
timed(lambda: torch.add(x, 1.0, out=y), 8 * n, "add   (read+write)")
