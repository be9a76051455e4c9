#!/usr/bin/env python3
"""python tools/h2d_probe.py -- what handing the train step HOST waveforms would cost (the bench's `value` is measured
with the waveforms resident in HBM; this is the PCIe-inclusive side note of DESIGN.md section 6).

  * H2D rate of one batch of waveforms (256 clips x 220 500 float32 = 226 MB) from pinned and from pageable memory;
  * the train step with the NEXT batch's upload issued on a copy stream at the start of every step (pinned memory,
    double-buffered device slots): the copy engine works beside the step's kernels;
  * the same with a blocking upload in front of every step (what a naive loop does).
"""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402
from bsed_amd.engine import FlatAdam, SEDTrainer  # noqa: E402
from bsed_amd.features import MelConfig, MelFrontEnd  # noqa: E402
from bsed_amd.models import CRNN, Predictor, weights_init  # noqa: E402

dev = torch.device("cuda:0")
B, sr = 256, 22050
n = 10 * sr
kw = dict(n_in_channel=1, nclass=20, attention=True, n_RNN_cell=128, n_layers_RNN=2, activation="glu", dropout=0.5,
          kernel_size=7 * [3], padding=7 * [1], stride=7 * [1], nb_filters=[16, 32, 64, 128, 128, 128, 128],
          pooling=[[2, 2], [2, 2], [1, 2], [1, 2], [1, 2], [1, 2], [1, 2]])
torch.manual_seed(2023)
mcfg = MelConfig(sr=sr)
fe = MelFrontEnd(mcfg)
T = fe.num_frames(n)
data = []
for seed in (2023, 4046):
    w, ev = bench.synth_waves(B, n, sr, seed, dev)
    data.append((w, bench.strong_labels(ev, T // 4, sr, mcfg.hop_size, 4, dev)))
host_pinned = [w.cpu().pin_memory() for w, _ in data]
host_paged = [w.cpu() for w, _ in data]
slots = [torch.empty_like(w) for w, _ in data]

for name, host in (("pinned", host_pinned), ("pageable", host_paged)):
    for _ in range(2):
        slots[0].copy_(host[0], non_blocking=True)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(5):
        slots[0].copy_(host[0], non_blocking=True)
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / 5
    print(f"H2D {name}: {host[0].numel() * 4 / dt / 1e9:.1f} GB/s, {dt * 1e3:.2f} ms per batch of {B} clips", flush=True)

crnn, pred = CRNN(**kw), Predictor(nclass=20, attention=True, n_RNN_cell=128)
weights_init(crnn); weights_init(pred)
tr = SEDTrainer(crnn, pred, optimizer=FlatAdam([crnn, pred], lr=1e-3), frontend=fe, seed=2023)
copy_stream = torch.cuda.Stream()
cnt = [0]


def step_resident():
    i = cnt[0]; cnt[0] += 1
    (w0, y0), (w1, _) = data[i % 2], data[(i + 1) % 2]
    return tr.train_step(w0, y0, from_wave=True, next_waves=(w1, None))


def step_async_upload():
    # slot (i+1) % 2 is transformed during this step (for step i+1): its upload was issued one step ago.  The slot of the
    # batch being trained now is free (its features were made during the previous step): refill it for step i+2.
    i = cnt[0]; cnt[0] += 1
    main = torch.cuda.current_stream()
    if tr._feat_stream is not None:
        copy_stream.wait_stream(tr._feat_stream)          # the previous transform of this slot has been enqueued there
    copy_stream.wait_stream(main)
    with torch.cuda.stream(copy_stream):
        slots[i % 2].copy_(host_pinned[i % 2], non_blocking=True)
    out = tr.train_step(slots[i % 2], data[i % 2][1], from_wave=True, next_waves=(slots[(i + 1) % 2], None))
    main.wait_stream(copy_stream)                         # next step's hook orders the feature stream after main
    return out


def step_blocking_upload():
    i = cnt[0]; cnt[0] += 1
    w = host_pinned[i % 2].to(dev)                        # on the step's stream, in front of the step
    return tr.train_step(w, data[i % 2][1], from_wave=True)


def run(name, step, K=30):
    cnt[0] = 0
    for _ in range(4):
        step()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(K):
        step()
    torch.cuda.synchronize()
    ms = (time.perf_counter() - t0) / K * 1e3
    print(f"{name}: {ms:.3f} ms/step = {B / ms * 1e3:.0f} clips/s", flush=True)


for k, s in enumerate(slots):
    s.copy_(data[k][0])
torch.cuda.synchronize()
for _ in range(2):
    run("waveforms resident in HBM (bench.py's value)", step_resident)
    run("pinned host waveforms, upload on a copy stream one step ahead", step_async_upload)
    run("pinned host waveforms, blocking upload in front of the step", step_blocking_upload)
