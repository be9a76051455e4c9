#!/usr/bin/env python3
"""Headline benchmark: 10 s clips/sec through the mel + CRNN train step (BASELINE.json metric).

    python bench.py --gpus N --steps K --warmup W            (N=1)
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...   (N>1)

One "step" = one pass of the hot path over one synthetic batch that is already resident in HBM as raw
waveforms: STFT/mel/dB on the GPU -> CRNN forward (dropout 0.5, train-mode BatchNorm) -> Predictor ->
BCE strong + BCE weak -> backward -> (RCCL all-reduce of the flat gradient arenas) -> Adam.  This is
BASELINE.json configs[2] ("main_baseline.py full CRNN train step on SYN, batch 256 per GPU"), the
configuration the metric is quoted on; data parallel = weak scaling (256 clips per GPU).

Prints ONE JSON line on rank 0 with `roofline` (the kernel with the largest total time of the step, every
launch HIP-event timed inside the timed region), `roofline_step` (whole-step fraction by SURVEY.md 8(d)'s formula)
and `cpu_baseline` (the CPU oracle timed on this box's host cores on a bounded sample, before the GPU is touched).
"""
import argparse
import json
import os
import re
import sys
import time

# hardware queues for the step's streams + RCCL's (bsed_amd/_lib.py sets the same default; here before torch is imported)
os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")

import numpy as np
import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

PEAK_FP32_MFMA_TFLOPS = 157.3   # /opt/skills/guides/MI355X_MICROARCH.md, "Peak FP32 (matrix)"
PEAK_BF16_MFMA_TFLOPS = 2500.0  # same guide, dense bf16 MFMA
PEAK_HBM_GBS = 8000.0


def synth_waves(B, n, sr, seed, device):
    """Deterministic synthetic clips (SURVEY.md 8d recipe, generated on the GPU): 0.1*N(0,1) floor plus three
    tones/chirps per clip with random onset/offset; returns (wave (B,n), events[(on,off,cls)] per clip)."""
    g = torch.Generator(device="cpu").manual_seed(seed)
    t = torch.arange(n, device=device, dtype=torch.float32) / sr
    idx = torch.arange(n, device=device)
    y = 0.1 * torch.randn((B, n), generator=torch.Generator(device=device).manual_seed(seed), device=device)
    dur = n / sr
    par = torch.rand((B, 3, 6), generator=g)
    ev = [[] for _ in range(B)]
    for e in range(3):   # one event per clip at a time, all clips at once (a few dozen launches instead of thousands)
        cols = {k: [] for k in ("f0", "f1", "amp", "on", "off", "i0", "i1")}
        for b in range(B):
            f0 = 500 + float(par[b, e, 0]) * (sr / 2 - 1000)
            f1 = 500 + float(par[b, e, 1]) * (sr / 2 - 1000) if float(par[b, e, 2]) < 0.5 else f0
            amp = 0.05 + 0.45 * float(par[b, e, 3])
            on = float(par[b, e, 4]) * (dur - 0.2)
            off = on + 0.2 + float(par[b, e, 5]) * (dur - on - 0.2)
            for k, v in (("f0", f0), ("f1", f1), ("amp", amp), ("on", on), ("off", off), ("i0", int(on * sr)),
                         ("i1", min(n, int(off * sr)))):
                cols[k].append(v)
            ev[b].append((on, off, int(par[b, e, 2] * 1e6) % 20))
        c = {k: torch.tensor(v, device=device, dtype=torch.int64 if k in ("i0", "i1") else torch.float32)[:, None]
             for k, v in cols.items()}
        for lo in range(0, B, 32):   # 32 clips at a time: the temporaries stay below 0.3 GB
            sl = slice(lo, min(B, lo + 32))
            tt = t[None, :] - c["on"][sl]
            chirp = c["amp"][sl] * torch.sin(2 * np.pi * (c["f0"][sl] * tt + 0.5 * (c["f1"][sl] - c["f0"][sl]) * tt * tt
                                                         / (c["off"][sl] - c["on"][sl])))
            y[sl] += chirp * ((idx[None, :] >= c["i0"][sl]) & (idx[None, :] < c["i1"][sl]))
    return y.clamp_(-1, 1), ev


def strong_labels(events, Tp, sr, hop, pooling, device):
    """frame index = int(t * sr // hop // pooling)  (reference ManyHotEncoder.py:121-122)"""
    y = torch.zeros((len(events), Tp, 20), dtype=torch.float32)
    for b, evb in enumerate(events):
        for on, off, c in evb:
            y[b, int(on * sr // hop // pooling):int(off * sr // hop // pooling), c] = 1
    return y.to(device)


def _cpu_mel_one(args):
    """process-pool worker of cpu_baseline (module level so that it pickles): numpy mel + dB of one synthetic clip"""
    i, sr, seconds = args
    from threadpoolctl import threadpool_limits
    from oracle import mel_oracle as mo
    c = mo.synth_clip(i, sr=sr, seconds=seconds)[0]
    T = 1 + len(c) // 255
    with threadpool_limits(1):       # one BLAS thread per worker: the pool (or the caller's loop) owns the parallelism
        return mo.transform_pair(mo.preprocess(c, sr=sr, fmax=min(16000.0, sr / 2)), T, unit_noise=np.zeros((T, 128)))[0]


def cpu_baseline(sr, seconds, threads):
    """The CPU oracle on this host's cores, as SURVEY.md 8(d) specifies it: the numpy mel restatement on 16 clips
    single-threaded and with a process pool, and the stock torch.nn CRNN train step (config 3) at B = 8 and B = 32 with
    1 warm-up + 3 timed steps on all threads.  Runs BEFORE this process touches the GPU (the pool's workers are fresh
    interpreters).  value = clips/s of pooled mel + the faster of the two train-step batch sizes."""
    import multiprocessing as mp
    from oracle import crnn_oracle as co
    from oracle import mel_oracle as mo
    n_mel = 16
    t0 = time.perf_counter()
    mels = [_cpu_mel_one((i, sr, seconds)) for i in range(n_mel)]
    t_mel_1 = (time.perf_counter() - t0) / n_mel
    workers = max(1, min(threads, n_mel))
    torch.set_num_threads(1)
    with mp.get_context("spawn").Pool(workers) as pool:
        pool.map(_cpu_mel_one, [(i, sr, 0.5) for i in range(workers)])          # start-up and imports, untimed
        t0 = time.perf_counter()
        pool.map(_cpu_mel_one, [(i, sr, seconds) for i in range(n_mel)], chunksize=1)
        t_mel_p = (time.perf_counter() - t0) / n_mel
    torch.set_num_threads(threads)
    steps = {}
    for B in (8, 32):
        crnn, pred = co.build(seed=1, dropout=0.5)
        crnn.train(); pred.train()
        opt = torch.optim.Adam(list(crnn.parameters()) + list(pred.parameters()), lr=1e-3)
        x = torch.from_numpy(np.stack([mels[i % n_mel] for i in range(B)]))
        Tp = x.shape[2] // 4
        y = torch.zeros((B, Tp, 20)); y[:, Tp // 3: Tp // 2, 3] = 1
        times = []
        for it in range(4):
            t0 = time.perf_counter()
            opt.zero_grad()
            loss, _ = co.train_losses(crnn, pred, x, y)
            loss.backward()
            opt.step()
            times.append(time.perf_counter() - t0)
        steps[B] = float(np.mean(times[1:])) / B
    t_step = min(steps.values())
    t_mel_p = min(t_mel_p, t_mel_1)
    return {"value": round(1.0 / (t_mel_p + t_step), 3), "unit": "clips/s", "cores": threads, "kind": "port",
            "mel_clips_per_s_single_thread": round(1.0 / t_mel_1, 2), "mel_clips_per_s_pool": round(1.0 / t_mel_p, 2),
            "train_step_clips_per_s": {f"B{b}": round(1.0 / v, 2) for b, v in steps.items()},
            "sample": f"numpy mel+dB on {n_mel} clips: 1 thread {t_mel_1*1e3:.0f} ms/clip, pool of {workers} processes "
                      f"{t_mel_p*1e3:.0f} ms/clip; torch CPU CRNN train step (oracle, fp32, {threads} threads), 1 warm-up "
                      f"+ 3 timed steps: B=8 {steps[8]*1e3:.0f} ms/clip, B=32 {steps[32]*1e3:.0f} ms/clip; {seconds:g} s "
                      f"clips @ {sr} Hz"}


SPLIT_KERNEL = re.compile(r"3[a-z]?_kernel")      # *3_kernel / *3n / *3p / *3s: split-fp32 operands on the bf16 cores
MFMA_KERNEL = re.compile(r"^(igemm|wgrad|glu_fwd3|glu_bwd3|glu_bwd_fused|gru_fwd_mfma|gru_bwd_mfma|glu16|b0_fwd|b0_bwd)")
ALGORITHMIC_MB_PER_CLIP = {22050: 128.9, 32000: 186.9}     # SURVEY.md 8(d), fp32 activations, mel stage included
ALGORITHMIC_MB_PER_CLIP_BF16 = {22050: 65.1, 32000: 94.4}  # SURVEY.md 8(d), bf16 activations (--dtype bf16)
B0_KERNEL = re.compile(r"^b0_(fwd|bwd)_kernel<")
# template instances of the split kernels whose ABF argument is 1: ONE bf16 MFMA per product.  ABF is the last argument,
# except in igemm3n_kernel<NWN, MW, STATS, PV, NT9, WPE, ABF[, SH]> (an eighth argument SH = 1 marks the 16 x 16 x 32 form)
class _SingleBf16:
    _last = re.compile(r"[<,] ?1>$")

    def search(self, name):
        m = re.match(r"^igemm3n_kernel<(.*)>$", name)
        if m:
            a = [t.strip() for t in m.group(1).split(",")]
            return True if len(a) >= 7 and a[6] == "1" else None
        return self._last.search(name)


SINGLE_BF16 = _SingleBf16()
STEP_GFLOP_PER_CLIP = {22050: 7.63, 32000: 11.05}           # SURVEY.md 8(d), train step = fwd + 2 x bwd


def kernel_roofline(name, launches, total_ms, flops_total, bytes_total):
    """roofline object of one kernel (template instance): which roof binds it and the achieved fraction.  The matrix
    ceiling of a split-fp32 kernel is the dense bf16 peak / 3 (three bf16 MFMAs per fp32 product); fp32-core MFMA kernels
    and plain VALU kernels share the 157.3 TFLOP/s fp32 ceiling."""
    split = SPLIT_KERNEL.search(name) is not None
    single = split and SINGLE_BF16.search(name.strip()) is not None    # the bf16 throughput mode's instances
    # ... and the first block's: b0_fwd / b0_bwd<.., 1> run their four 16 x 16 x 16 contractions as single bf16 MFMAs
    single = single or (B0_KERNEL.search(name) is not None and SINGLE_BF16.search(name.strip()) is not None)
    peak_tf = PEAK_BF16_MFMA_TFLOPS if single else (PEAK_BF16_MFMA_TFLOPS / 3.0 if split else PEAK_FP32_MFMA_TFLOPS)
    avg_ms = total_ms / launches
    tflops = flops_total / (total_ms * 1e-3) / 1e12
    gbs = bytes_total / (total_ms * 1e-3) / 1e9
    intensity = flops_total / max(bytes_total, 1.0)
    ridge = peak_tf * 1e12 / (PEAK_HBM_GBS * 1e9)
    on_matrix = MFMA_KERNEL.search(name) is not None
    basis = ("dense bf16 MFMA 2500 (bf16 activations, one MFMA per product)" if single else
             "dense bf16 MFMA 2500 / 3 (bf16x3 split-fp32 operands)" if split else
             "fp32 MFMA 157.3 (v_mfma_f32_32x32x2_f32 / 16x16x4_f32, exact fp32)" if on_matrix else "fp32 vector 157.3 (no matrix-core work in this kernel)")
    r = {"traffic": None, "kernel": name, "avg_launch_ms": round(avg_ms, 4), "launches": launches,
         "algorithmic_gflop_per_launch": round(flops_total / launches / 1e9, 3),
         "algorithmic_mbytes_per_launch": round(bytes_total / launches / 1e6, 2),
         "intensity_flop_per_byte": round(intensity, 1), "ridge_flop_per_byte": round(ridge, 1),
         "compute_tflops": round(tflops, 2), "compute_peak_tflops": round(peak_tf, 1), "compute_peak_basis": basis,
         "hbm_gbs": round(gbs, 1)}
    if intensity >= ridge and on_matrix:
        r.update({"bound": "mfma", "achieved": round(tflops, 3), "peak": round(peak_tf, 1), "unit": "TFLOP/s",
                  "frac": round(tflops / peak_tf, 4)})
    else:
        r.update({"bound": "hbm", "achieved": round(gbs, 1), "peak": PEAK_HBM_GBS, "unit": "GB/s",
                  "frac": round(gbs / PEAK_HBM_GBS, 4)})
        if intensity >= ridge:   # a VALU kernel above the ridge: the fp32 vector ceiling is the tighter roof
            r["valu_frac"] = round(tflops / peak_tf, 4)
    return r


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=40)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--batch", type=int, default=None, help="clips per GPU (default 256; 64 for --mode cnn)")
    ap.add_argument("--sr", type=int, default=22050, help="22050 = BASELINE measurement config, 32000 = reference config")
    ap.add_argument("--seconds", type=float, default=10.0)
    ap.add_argument("--mode", choices=["crnn", "mt", "ada", "cnn"], default="crnn",
                    help="crnn = BASELINE configs[2] (the headline metric); mt = configs[3] (student + EMA teacher + "
                         "consistency, half the batch synthetic, half real); ada = configs[4] (domain-adversarial head); "
                         "cnn = configs[1] (CNN-only tagging forward, CRNN_pred, batch 64)")
    ap.add_argument("--dtype", choices=["f32", "bf16"], default="f32",
                    help="f32 (default, the parity headline): fp32 tensors, split-fp32 contractions; bf16: the throughput mode "
                         "BASELINE configs[1-2] name -- bf16 CNN activations in HBM, one bf16 MFMA per product, fp32 accumulation "
                         "/ statistics / master weights / optimizer (tolerances: tests/test_bf16_mode_gpu.py)")
    ap.add_argument("--graph", action="store_true",
                    help="--mode crnn: capture the step in a HIP graph (SEDTrainer.capture_step) and time replays; the "
                         "per-launch kernel timer is off (events are not captured).  For the reference's batch: --batch 24")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-pipeline", action="store_true",
                    help="mel transform inside the step instead of one step ahead on the feature stream")
    ap.add_argument("--no-kernel-timer", action="store_true")
    ap.add_argument("--host-waves", action="store_true",
                    help="diagnostic: waveform batches in pinned HOST memory, uploaded one step ahead on the trainer's copy "
                         "stream (PCIe-inclusive rate; the headline keeps its inputs resident in HBM)")
    ap.add_argument("--dropout", type=float, default=0.5,
                    help="dropout of the CRNN (0.5 = the reference's crnn_kwargs and the BASELINE configuration; other values "
                         "are diagnostic: 0 shows what the mask hashes and the final dropout kernels cost)")
    ap.add_argument("--timer-steps", type=int, default=4,
                    help="the per-launch HIP events (roofline leg) are recorded during the first N of the timed steps "
                         "(0 = all of them): an evented step is ~0.3 ms (2 %%) longer -- measured at --steps 20: 13.54 ms "
                         "without events, 13.59 / 13.63 / 13.71 with 3 / 5 / 10 evented steps; the kernel averages agree")
    args = ap.parse_args()
    if args.batch is None:
        args.batch = 64 if args.mode == "cnn" else 256
    if args.dtype == "bf16":
        if args.mode in ("ada",):
            raise SystemExit("--dtype bf16 covers --mode crnn / mt / cnn (the discriminator has no bf16-activation path)")
        os.environ["BSED_CONV_MODE"] = "bf16"      # read by CRNN.__init__

    t_start = time.perf_counter()
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if args.gpus != world:
        if world == 1 and args.gpus > 1:
            raise SystemExit("launch with torch.distributed.run --nproc-per-node N for --gpus N")

    def log(msg):
        if rank == 0:
            print(f"[bench +{time.perf_counter() - t_start:7.1f}s] {msg}", file=sys.stderr, flush=True)

    # CPU baseline first, on rank 0 at N = 1 only, BEFORE this process initialises the GPU: its process pool starts
    # fresh interpreters, which a GPU-initialised parent must not do on the GPU boxes
    cpu = None
    if not args.no_cpu_baseline and world == 1 and args.mode == "crnn":
        try:
            ncpu = len(os.sched_getaffinity(0))
        except AttributeError:
            ncpu = os.cpu_count() or 1
        cpu = cpu_baseline(args.sr, args.seconds, max(1, min(ncpu, 16)))
        log(f"cpu baseline done: {cpu['value']} clips/s on {cpu['cores']} threads")

    ngpu = torch.cuda.device_count()
    ranks_seen = 1
    # one process per GPU; BSED_DIST_BACKEND=gloo lets several ranks share one card for rehearsals of the N>1 path
    backend = os.environ.get("BSED_DIST_BACKEND", "nccl")
    dev_index = local_rank % max(ngpu, 1) if backend != "nccl" else local_rank
    torch.cuda.set_device(dev_index)
    dev = torch.device("cuda", dev_index)
    if "RANK" in os.environ:
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        # RCCL prints its version banner (NCCL_DEBUG=VERSION on the GPU boxes) to stdout when the communicator is
        # created: create it here, under a temporary stdout -> stderr redirection, so that stdout carries the ONE JSON
        # line and nothing else
        sys.stdout.flush()
        saved_fd = os.dup(1)
        os.dup2(2, 1)
        try:
            if backend == "nccl":
                torch.distributed.init_process_group("nccl", device_id=dev)
            else:
                torch.distributed.init_process_group(backend)
            probe = torch.ones(1, device=dev)
            torch.distributed.all_reduce(probe)
            torch.cuda.synchronize()
            ranks_seen = int(probe.item())       # evidence in the JSON line that the backend summed over N ranks
        finally:
            os.dup2(saved_fd, 1)
            os.close(saved_fd)

    from bsed_amd import ops
    from bsed_amd.engine import FlatAdam, FlatSGD, SEDTrainer
    from bsed_amd.features import MelConfig, MelFrontEnd
    from bsed_amd.models import CRNN, CRNN_pred, Predictor, weights_init

    kw = dict(n_in_channel=1, nclass=20, attention=True, n_RNN_cell=128, n_layers_RNN=2, activation="glu",
              dropout=args.dropout, kernel_size=7 * [3], padding=7 * [1], stride=7 * [1],
              nb_filters=[16, 32, 64, 128, 128, 128, 128],
              pooling=[[2, 2], [2, 2], [1, 2], [1, 2], [1, 2], [1, 2], [1, 2]])
    torch.manual_seed(2023)
    mcfg = MelConfig(sr=args.sr)
    fe = MelFrontEnd(mcfg)
    B, n = args.batch, int(args.seconds * args.sr)
    wav, ev = synth_waves(B, n, args.sr, 2023 + rank, dev)
    T = fe.num_frames(n)
    Tp = T // 4
    tr = None
    if args.mode == "cnn":
        # BASELINE configs[1]: the reference's CNN-only tagger (CRNN_pred, models/CRNN_GRL.py:206-290) in eval mode:
        # waveform -> mel/dB -> 7 conv blocks -> sigmoid features + class-softmax attention pooling
        kwp = dict(kw); kwp.update(nclass=128, n_RNN_cell=64)
        tagger = CRNN_pred(**kwp)
        weights_init(tagger)
        tagger.eval()

        def step():
            with torch.no_grad():
                return tagger(fe.transform(wav, max_frames=T))
    else:
        crnn, pred = CRNN(**kw), Predictor(nclass=20, attention=True, n_RNN_cell=128)
        weights_init(crnn); weights_init(pred)
        extra = {}
        optimizer = FlatAdam([crnn, pred], lr=1e-3)
        if args.mode == "mt":
            # reference src/main_scmt.py: the teacher is a second CRNN/Predictor pair that only ever receives the EMA
            ema_c, ema_p = CRNN(**kw), Predictor(nclass=20, attention=True, n_RNN_cell=128)
            ema_c.load_state_dict(crnn.state_dict()); ema_p.load_state_dict(pred.state_dict())
            extra = dict(ema_crnn=ema_c, ema_predictor=ema_p)
        elif args.mode == "ada":
            from bsed_amd.disc import Clip_Discriminator, ConditionalDomainAdversarialLoss
            disc = Clip_Discriminator()
            # reference src/main_scmt_ada_weak.py:854-866: SGD-Nesterov(momentum 0.9, weight decay 1e-4) for BOTH optimizers
            optimizer = FlatSGD([crnn, pred], lr=1e-3, momentum=0.9, weight_decay=1e-4, nesterov=True)
            extra = dict(domain_loss=ConditionalDomainAdversarialLoss(disc),
                         optimizer_d=FlatSGD([disc], lr=1e-4, momentum=0.9, weight_decay=1e-4, nesterov=True))
        tr = SEDTrainer(crnn, pred, optimizer=optimizer, frontend=fe, seed=2023, **extra)
        tr.broadcast_parameters()
        y = strong_labels(ev, Tp, args.sr, mcfg.hop_size, 4, dev)
        # Two-deep input pipeline (SEDTrainer.train_step(next_waves=...)): the steps alternate between TWO resident
        # synthetic batches, and each step enqueues the mel transform of the batch the NEXT step trains on (on the
        # feature stream, beside this step's recurrences).  Every step still transforms one batch -- one step ahead --
        # and every step trains on a batch other than the previous step's.  --no-pipeline: transform inside the step.
        wav2, ev2 = synth_waves(B, n, args.sr, 4046 + rank, dev)
        y2 = strong_labels(ev2, Tp, args.sr, mcfg.hop_size, 4, dev)
        count = [0]
        if args.mode == "crnn":
            batches = [(wav, y), (wav2, y2)]

            def step():
                (w0, y0), (w1, _) = batches[count[0] % 2], batches[(count[0] + 1) % 2]
                count[0] += 1
                return tr.train_step(w0, y0, from_wave=True, next_waves=None if args.no_pipeline else (w1, None))
        else:
            # half the clips play the synthetic (strongly labelled) batch, half the real batch (weak labels for mt)
            h = B // 2
            batches = []
            for w_, y_ in ((wav, y), (wav2, y2)):
                batches.append((w_[:h].contiguous(), y_[:h].contiguous(), w_[h:].contiguous(),
                                y_[h:].max(1)[0].contiguous() if args.mode == "mt" else None))
            del wav2

            def step():
                b0, b1 = batches[count[0] % 2], batches[(count[0] + 1) % 2]
                count[0] += 1
                return tr.train_step(b0[0], b0[1], b0[2], b0[3], from_wave=True,
                                     next_waves=None if args.no_pipeline else (b1[0], b1[2]))

    if args.host_waves and tr is not None:
        # diagnostic: the caller's batches live in PINNED HOST memory; SEDTrainer uploads the next step's waveforms on
        # its copy stream at the start of every step (PCIe-inclusive rate: never the headline `value`)
        if args.mode == "crnn":
            batches = [(w_.cpu().pin_memory(), y_) for w_, y_ in batches]
        else:
            batches = [(b[0].cpu().pin_memory(), b[1], b[2].cpu().pin_memory(), b[3]) for b in batches]
    if tr is not None:
        tr.arena.timing = world > 1 or tr.arena.exchange_single_rank
    log(f"data ready: B={B} n={n} T={T} Tp={Tp}")
    use_timer = not args.no_kernel_timer and rank == 0 and not args.graph
    if args.graph:
        if args.mode != "crnn" or world != 1:
            raise SystemExit("--graph covers --mode crnn on one rank")
        # the captured step transforms its own waveform batch (no cross-step feature pipeline inside a graph)
        tr.capture_step(batches[0][0], batches[0][1], from_wave=True, warmup=max(args.warmup, 2))

        def step():   # noqa: F811
            w0, y0 = batches[count[0] % 2]
            count[0] += 1
            return tr.replay_step(w0, y0)
    # The warm-up runs with a throw-away kernel timer: the first few hundred HIP events of a process make the runtime
    # grow its signal pool (a one-time ~45 ms stall, measured on a fresh box), which must not land in the timed region.
    if use_timer:
        ops.KernelTimer.prime(2 * 400 * (args.steps + args.warmup) + 512)
        ops.set_timer(ops.KernelTimer())
    for i in range(args.warmup):
        step()
        torch.cuda.synchronize()
        log(f"warmup step {i} done")
    timer = None
    if use_timer:
        # (the mean-teacher step runs on a high-priority stream of the trainer: that is the step's "main" stream)
        own = getattr(tr, "_step_stream", None) if tr is not None else None
        timer = ops.KernelTimer(main_stream=own.cuda_stream) if own is not None else ops.KernelTimer()
        ops.set_timer(timer)
    torch.cuda.synchronize()
    if world > 1:
        torch.distributed.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    tsteps = args.steps if args.timer_steps <= 0 else min(args.timer_steps, args.steps)
    for i in range(args.steps):
        if timer is not None and i == tsteps:
            ops.set_timer(None)     # the remaining timed steps run without per-launch events
        out = step()
    host_enqueue = time.perf_counter() - t0     # the host is done enqueueing; the GPU may still be running
    torch.cuda.synchronize()
    if world > 1:
        torch.distributed.barrier()
    torch.cuda.synchronize()
    elapsed = time.perf_counter() - t0
    ops.set_timer(None)
    log(f"timed region done: {elapsed:.3f}s for {args.steps} steps (host enqueue {1e3 * host_enqueue / args.steps:.2f} ms/step)")
    if world > 1:
        tmax = torch.tensor([elapsed], device=dev, dtype=torch.float64)
        torch.distributed.all_reduce(tmax, op=torch.distributed.ReduceOp.MAX)
        elapsed = float(tmax)
    dp = None
    if tr is not None and (world > 1 or tr.arena.exchange_single_rank):
        # what the gradient exchange cost: (a) how long the step's stream actually WAITED in GradArena.finish() for the
        # early + tail all-reduces (events around the waits, inside the timed region: the exposed part of the exchange);
        # (b) the two all-reduces alone, back to back on an otherwise idle GPU, after the timed region (every rank takes
        # part: collective).  All ranks run this block.
        ar = tr.arena
        waits = [a.elapsed_time(b) for a, b in ar.wait_events]
        alone = {}
        for nm, buf in (("early", ar.early), ("tail", ar.tail)):
            if buf.numel() == 0:
                continue
            scratch = torch.zeros_like(buf)
            for _ in range(3):
                torch.distributed.all_reduce(scratch)
            torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(20):
                torch.distributed.all_reduce(scratch)
            e1.record()
            torch.cuda.synchronize()
            alone[nm] = e0.elapsed_time(e1) / 20
        dp = {"ranks_seen": ranks_seen, "backend": backend, "allreduce_mb": round(ar.flat.numel() * 4 / 1e6, 3),
              "early_mb": round(ar.early.numel() * 4 / 1e6, 3), "tail_kb": round(ar.tail.numel() * 4 / 1e3, 2),
              "exchanges_per_step": round(ar.exchanges / max(1, args.steps + args.warmup), 2),
              "finish_wait_ms_per_step": round(float(np.mean(waits)), 4) if waits else None,
              "finish_wait_ms_max": round(float(np.max(waits)), 4) if waits else None,
              "allreduce_early_ms_alone": round(alone.get("early", 0.0), 4),
              "allreduce_tail_ms_alone": round(alone.get("tail", 0.0), 4)}
    if args.mode == "cnn":
        loss = float(out[1].mean())      # mean weak probability: a finite-output check, not a loss
    else:
        loss = SEDTrainer.loss_value(out)
    if rank != 0:
        if torch.distributed.is_initialized():
            torch.distributed.destroy_process_group()
        return

    value = world * B * args.steps / elapsed
    roofline = None
    extra_fields = {}
    if timer is not None:
        # {(kernel, shape tag): (launches, total_ms, avg_ms, flops_per_launch, algorithmic_bytes_per_launch)}
        summ = timer.summary()
        tot_ms = sum(v[1] for v in summ.values())
        for k in sorted(summ, key=lambda k: -summ[k][1]):
            c, tms, ams, fl, nb = summ[k]
            log("  %-58s x%-3d %8.3f ms/step  avg %7.3f ms  %6.1f TFLOP/s %6.0f GB/s" % (
                "%s %s" % k, c // tsteps, tms / tsteps, ams, fl / ams / 1e9, nb / ams / 1e6))
        # the roofline object is quoted per KERNEL (template instance, the unit rocprofv3 --stats aggregates on): all its
        # launches of the timed region with their algorithmic FLOPs and bytes; EVERY launch of the step is timed (mel,
        # first block, BatchNorm, GRU, head, optimizer and glue included), so the kernel named is the step's dominant one
        byk = {}
        for k, (c, tms, ams, fl, nb) in summ.items():
            a = byk.setdefault(k[0], [0, 0.0, 0.0, 0.0])
            a[0] += c; a[1] += tms; a[2] += fl * c; a[3] += nb * c
        for name in sorted(byk, key=lambda n: -byk[n][1]):
            c, tms, fl, nb = byk[name]
            log("  KERNEL %-40s x%-3d %8.3f ms/step  avg %7.3f ms  %6.1f TFLOP/s %6.0f GB/s" % (
                name, c // tsteps, tms / tsteps, tms / c, fl / tms / 1e9, nb / tms / 1e6))
        # launches on other streams (GRU weight gradients beside the next recurrence; the next batch's mel transform
        # beside the GRU forward) overlap main-stream kernels: their durations include contention and are listed apart;
        # their time is inside the step, not in kernels_ms_per_step
        sside = timer.summary(side=True)
        byks = {}
        for k, (c, tms, ams, fl, nb) in sside.items():
            a = byks.setdefault(k[0], [0, 0.0, 0.0, 0.0])
            a[0] += c; a[1] += tms; a[2] += fl * c; a[3] += nb * c
        for nm in sorted(byks, key=lambda n: -byks[n][1]):
            c, tms, fl, nb = byks[nm]
            log("  SIDE-STREAM KERNEL (overlapped) %-28s x%-3d %8.3f ms/step  avg %7.3f ms  %6.1f TFLOP/s %6.0f GB/s" % (
                nm, c // tsteps, tms / tsteps, tms / c, fl / tms / 1e9, nb / tms / 1e6))
        extra_side = sum(v[0] for v in byks.values()) // tsteps
        name = max(byk, key=lambda n: byk[n][1])
        roofline = kernel_roofline(name, *byk[name])
        roofline["side_stream_launches_not_timed"] = extra_side
        roofline["share_of_kernel_time"] = round(byk[name][1] / tot_ms, 3)
        roofline["kernels_ms_per_step"] = round(tot_ms / tsteps, 3)
        roofline["timed_steps_with_events"] = tsteps
        # HBM bytes per launch of that kernel from the PMC passes (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE, separate
        # passes, FETCH doubled as MI355X_MICROARCH.md prescribes for gfx950); only valid for the profiled workload
        try:
            if args.mode == "crnn" and B == 256 and args.sr == 22050 and args.seconds == 10.0:
                for fn in (("r03_pmc_traffic_bf16.json",) if args.dtype == "bf16" else ()) + \
                          ("r03_pmc_traffic.json", "r02_pmc_traffic.json", "r01_pmc_traffic.json"):
                    path = os.path.join(ROOT, "profiles", fn)
                    if not os.path.exists(path):
                        continue
                    pmc = json.load(open(path))["kernels"].get(name)
                    if pmc:
                        roofline["traffic"] = round(pmc["hbm_bytes_per_launch"])
                        roofline["traffic_source"] = "profiles/" + fn
                        break
        except (OSError, KeyError, ValueError):
            pass
        # the bf16-MFMA-bound (split-fp32) kernel with the largest total, for continuity with round 1's line
        mf = [nm for nm in byk if MFMA_KERNEL.search(nm) and SPLIT_KERNEL.search(nm) and
              byk[nm][2] / max(byk[nm][3], 1.0) >= (PEAK_BF16_MFMA_TFLOPS / 3 if SPLIT_KERNEL.search(nm) else
                                                    PEAK_FP32_MFMA_TFLOPS) * 1e12 / (PEAK_HBM_GBS * 1e9)]
        if mf:
            nm = max(mf, key=lambda n: byk[n][1])
            extra_fields["roofline_top_mfma_kernel"] = {k: v for k, v in kernel_roofline(nm, *byk[nm]).items()
                                                        if k in ("kernel", "bound", "achieved", "peak", "unit", "frac",
                                                                 "avg_launch_ms", "launches")}
    if args.mode != "cnn" and args.sr in ALGORITHMIC_MB_PER_CLIP:
        # whole-step fractions by SURVEY.md 8(d)'s formula: clips/s/GPU x algorithmic bytes (FLOPs) per clip over the peak
        per_gpu = value / world
        scale = {"crnn": 1.0, "mt": None, "ada": None}[args.mode]
        if scale is not None:
            mbt = ALGORITHMIC_MB_PER_CLIP_BF16 if args.dtype == "bf16" else ALGORITHMIC_MB_PER_CLIP
            mb, gf = mbt[args.sr] * args.seconds / 10.0, STEP_GFLOP_PER_CLIP[args.sr] * args.seconds / 10.0
            extra_fields["roofline_step"] = {
                "bound": "hbm", "algorithmic_mb_per_clip": round(mb, 1), "achieved": round(per_gpu * mb / 1e3, 1),
                "peak": PEAK_HBM_GBS, "unit": "GB/s", "frac": round(per_gpu * mb * 1e6 / (PEAK_HBM_GBS * 1e9), 4),
                "step_gflop_per_clip": gf, "tflops": round(per_gpu * gf / 1e3, 1),
                ("mfma_frac_bf16" if args.dtype == "bf16" else "mfma_frac_bf16x3"):
                    round(per_gpu * gf * 1e9 / (PEAK_BF16_MFMA_TFLOPS / (1 if args.dtype == "bf16" else 3) * 1e12), 4)}
    workload = {
        "crnn": "waveform->STFT/mel/dB->CRNN(7 conv/BN/GLU/pool + 2xBiGRU128)->Predictor->BCE strong+weak"
                "->backward->Adam; BASELINE configs[2] (main_baseline.py train step on SYN)",
        "mt": "mean teacher: student on B/2 synthetic + B/2 real clips, EMA teacher forward on the noisy real half, "
              "consistency MSE, backward, Adam, EMA update; BASELINE configs[3] (main_scmt.py)",
        "ada": "domain-adversarial: student on B/2 synthetic + B/2 real clips, Clip_Discriminator + gradient reverse "
               "on both encodings, backward, SGD-Nesterov(0.9, wd 1e-4) on CRNN+Predictor and on the discriminator; "
               "BASELINE configs[4] (main_scmt_ada_weak.py)",
        "cnn": "waveform->STFT/mel/dB->7 conv/BN/GLU/pool blocks->sigmoid features + class-softmax attention pooling, "
               "eval-mode forward only; BASELINE configs[1] (CNN-only tagging forward: the reference's CRNN_pred)",
    }[args.mode]
    line = {
        "metric": "10 s clips/sec through mel+CRNN train step" if args.mode != "cnn"
                  else "10 s clips/sec through mel+CNN tagging forward",
        "value": round(value, 2), "unit": "clips/s",
        "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
        "ms_per_step": round(1e3 * elapsed / args.steps, 3), "higher_is_better": True, "scaling": "weak",
        "vs_baseline": None, "dtype": args.dtype, "data": "synthetic",
        "contraction_mode": os.environ.get("BSED_CONV_MODE", "bf16x3"),
        "config": {"workload": workload, "mode": args.mode,
                   "clip_seconds": args.seconds, "sr": args.sr, "frames": T, "out_frames": Tp,
                   "batch_per_gpu": B, "global_batch": B * world, "parallelism": f"dp{world}",
                   "dropout": 0.0 if args.mode == "cnn" else args.dropout,
                   "hip_graph": bool(args.graph),
                   "input_pipeline": "none" if (args.mode == "cnn" or args.no_pipeline or args.graph) else
                   "2 alternating resident batches; each step transforms the next step's waveforms (feature stream)"},
        "roofline": roofline, "cpu_baseline": cpu, "final_loss": round(loss, 5),
        "ranks_seen": ranks_seen, "data_parallel": dp,
    }
    if args.host_waves:
        line["config"]["input_pipeline"] = ("pinned HOST waveforms, uploaded one step ahead on the trainer's copy stream "
                                            "(PCIe-inclusive diagnostic, not the headline)")
    line.update(extra_fields)
    print(json.dumps(line))
    if torch.distributed.is_initialized():
        torch.distributed.destroy_process_group()


if __name__ == "__main__":
    main()
