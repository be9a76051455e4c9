#!/usr/bin/env python3
"""Headline benchmark: 10 s clips/sec through the mel + CRNN train step (BASELINE.json metric).

    python bench.py --gpus N --steps K --warmup W            (N=1)
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...   (N>1)

One "step" = one pass of the hot path over one synthetic batch that is already resident in HBM as raw
waveforms: STFT/mel/dB on the GPU -> CRNN forward (dropout 0.5, train-mode BatchNorm) -> Predictor ->
BCE strong + BCE weak -> backward -> (RCCL all-reduce of the flat gradient arenas) -> Adam.  This is
BASELINE.json configs[2] ("main_baseline.py full CRNN train step on SYN, batch 256 per GPU"), the
configuration the metric is quoted on; data parallel = weak scaling (256 clips per GPU).

Prints ONE JSON line on rank 0 with `roofline` (dominant MFMA kernel, HIP-event timed inside the timed
region) and `cpu_baseline` (the CPU oracle timed on this box's host cores on a bounded sample).
"""
import argparse
import json
import os
import re
import sys
import time

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
    y = 0.1 * torch.randn((B, n), generator=torch.Generator(device=device).manual_seed(seed), device=device)
    dur = n / sr
    ev = []
    par = torch.rand((B, 3, 6), generator=g)
    for b in range(B):
        evb = []
        for e in range(3):
            f0 = 500 + float(par[b, e, 0]) * (sr / 2 - 1000)
            f1 = 500 + float(par[b, e, 1]) * (sr / 2 - 1000) if float(par[b, e, 2]) < 0.5 else f0
            amp = 0.05 + 0.45 * float(par[b, e, 3])
            on = float(par[b, e, 4]) * (dur - 0.2)
            off = on + 0.2 + float(par[b, e, 5]) * (dur - on - 0.2)
            i0, i1 = int(on * sr), min(n, int(off * sr))
            tt = t[i0:i1] - on
            y[b, i0:i1] += amp * torch.sin(2 * np.pi * (f0 * tt + 0.5 * (f1 - f0) * tt * tt / (off - on)))
            evb.append((on, off, int(par[b, e, 2] * 1e6) % 20))
        ev.append(evb)
    return y.clamp_(-1, 1), ev


def strong_labels(events, Tp, sr, hop, pooling, device):
    """frame index = int(t * sr // hop // pooling)  (reference ManyHotEncoder.py:121-122)"""
    y = torch.zeros((len(events), Tp, 20), dtype=torch.float32)
    for b, evb in enumerate(events):
        for on, off, c in evb:
            y[b, int(on * sr // hop // pooling):int(off * sr // hop // pooling), c] = 1
    return y.to(device)


def cpu_baseline(sr, seconds, threads):
    """The CPU oracle (numpy mel restatement + stock torch.nn CRNN train step) on this host's cores."""
    from oracle import crnn_oracle as co
    from oracle import mel_oracle as mo
    torch.set_num_threads(threads)
    n_mel = 4
    clips = [mo.synth_clip(i, sr=sr, seconds=seconds)[0] for i in range(n_mel)]
    t0 = time.perf_counter()
    mels = [mo.transform_pair(mo.preprocess(c, sr=sr, fmax=min(16000.0, sr / 2)), 1 + len(c) // 255,
                              unit_noise=np.zeros((1 + len(c) // 255, 128)))[0] for c in clips]
    t_mel = (time.perf_counter() - t0) / n_mel
    B = 8
    crnn, pred = co.build(seed=1, dropout=0.5)
    crnn.train(); pred.train()
    opt = torch.optim.Adam(list(crnn.parameters()) + list(pred.parameters()), lr=1e-3)
    x = torch.from_numpy(np.stack([mels[i % n_mel] for i in range(B)]))
    Tp = x.shape[2] // 4
    y = torch.zeros((B, Tp, 20)); y[:, Tp // 3: Tp // 2, 3] = 1
    times = []
    for it in range(3):
        t0 = time.perf_counter()
        opt.zero_grad()
        loss, _ = co.train_losses(crnn, pred, x, y)
        loss.backward()
        opt.step()
        times.append(time.perf_counter() - t0)
    t_step = float(np.mean(times[1:])) / B
    return {"value": 1.0 / (t_mel + t_step), "unit": "clips/s", "cores": threads, "kind": "port",
            "sample": f"numpy mel on {n_mel} clips (single thread, {t_mel*1e3:.0f} ms/clip) + torch CPU CRNN train "
                      f"step B={B}, 1 warm-up + 2 timed ({t_step*1e3:.0f} ms/clip), {seconds:g} s clips @ {sr} Hz"}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=8)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--batch", type=int, default=256, help="clips per GPU")
    ap.add_argument("--sr", type=int, default=22050, help="22050 = BASELINE measurement config, 32000 = reference config")
    ap.add_argument("--seconds", type=float, default=10.0)
    ap.add_argument("--mode", choices=["crnn", "mt", "ada"], default="crnn",
                    help="crnn = BASELINE configs[2] (the headline metric); mt = configs[3] (student + EMA teacher + "
                         "consistency, half the batch synthetic, half real); ada = configs[4] (domain-adversarial head)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-kernel-timer", action="store_true")
    args = ap.parse_args()

    t_start = time.perf_counter()
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if args.gpus != world:
        if world == 1 and args.gpus > 1:
            raise SystemExit("launch with torch.distributed.run --nproc-per-node N for --gpus N")
    ngpu = torch.cuda.device_count()
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
        finally:
            os.dup2(saved_fd, 1)
            os.close(saved_fd)

    from bsed_amd import ops
    from bsed_amd.engine import FlatAdam, SEDTrainer
    from bsed_amd.features import MelConfig, MelFrontEnd
    from bsed_amd.models import CRNN, Predictor, weights_init

    kw = dict(n_in_channel=1, nclass=20, attention=True, n_RNN_cell=128, n_layers_RNN=2, activation="glu",
              dropout=0.5, kernel_size=7 * [3], padding=7 * [1], stride=7 * [1],
              nb_filters=[16, 32, 64, 128, 128, 128, 128],
              pooling=[[2, 2], [2, 2], [1, 2], [1, 2], [1, 2], [1, 2], [1, 2]])
    torch.manual_seed(2023)
    crnn, pred = CRNN(**kw), Predictor(nclass=20, attention=True, n_RNN_cell=128)
    weights_init(crnn); weights_init(pred)
    mcfg = MelConfig(sr=args.sr)
    fe = MelFrontEnd(mcfg)
    extra = {}
    if args.mode == "mt":
        # reference src/main_scmt.py: the teacher is a second CRNN/Predictor pair that only ever receives the EMA
        ema_c, ema_p = CRNN(**kw), Predictor(nclass=20, attention=True, n_RNN_cell=128)
        ema_c.load_state_dict(crnn.state_dict()); ema_p.load_state_dict(pred.state_dict())
        extra = dict(ema_crnn=ema_c, ema_predictor=ema_p)
    elif args.mode == "ada":
        from bsed_amd.disc import Clip_Discriminator, ConditionalDomainAdversarialLoss
        from bsed_amd.engine import FlatSGD
        disc = Clip_Discriminator()
        extra = dict(domain_loss=ConditionalDomainAdversarialLoss(disc),
                     optimizer_d=FlatSGD([disc], lr=1e-4, momentum=0.9, weight_decay=0.0))
    tr = SEDTrainer(crnn, pred, optimizer=FlatAdam([crnn, pred], lr=1e-3), frontend=fe, seed=2023, **extra)
    tr.broadcast_parameters()

    B, n = args.batch, int(args.seconds * args.sr)
    wav, ev = synth_waves(B, n, args.sr, 2023 + rank, dev)
    T = fe.num_frames(n)
    Tp = T // 4
    y = strong_labels(ev, Tp, args.sr, mcfg.hop_size, 4, dev)

    if args.mode == "crnn":
        def step():
            return tr.train_step(wav, y, from_wave=True)
    else:
        # half the clips play the synthetic (strongly labelled) batch, half the real batch (weak labels for mt)
        h = B // 2
        wav_s, y_s, wav_r = wav[:h].contiguous(), y[:h].contiguous(), wav[h:].contiguous()
        yw_r = y[h:].max(1)[0].contiguous() if args.mode == "mt" else None

        def step():
            return tr.train_step(wav_s, y_s, wav_r, yw_r, from_wave=True)

    def log(msg):
        if rank == 0:
            print(f"[bench +{time.perf_counter() - t_start:7.1f}s] {msg}", file=sys.stderr, flush=True)

    log(f"data ready: B={B} n={n} T={T} Tp={Tp}")
    use_timer = not args.no_kernel_timer and rank == 0
    # The warm-up runs with a throw-away kernel timer: the first few hundred HIP events of a process make the runtime
    # grow its signal pool (a one-time ~45 ms stall, measured on a fresh box), which must not land in the timed region.
    if use_timer:
        ops.KernelTimer.prime(2 * 120 * (args.steps + args.warmup) + 512)
        ops.set_timer(ops.KernelTimer())
    for i in range(args.warmup):
        step()
        torch.cuda.synchronize()
        log(f"warmup step {i} done")
    timer = None
    if use_timer:
        timer = ops.KernelTimer()
        ops.set_timer(timer)
    torch.cuda.synchronize()
    if world > 1:
        torch.distributed.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        out = step()
    torch.cuda.synchronize()
    if world > 1:
        torch.distributed.barrier()
    torch.cuda.synchronize()
    elapsed = time.perf_counter() - t0
    ops.set_timer(None)
    log(f"timed region done: {elapsed:.3f}s for {args.steps} steps")
    if world > 1:
        tmax = torch.tensor([elapsed], device=dev, dtype=torch.float64)
        torch.distributed.all_reduce(tmax, op=torch.distributed.ReduceOp.MAX)
        elapsed = float(tmax)
    loss = SEDTrainer.loss_value(out)
    if rank != 0:
        if torch.distributed.is_initialized():
            torch.distributed.destroy_process_group()
        return

    value = world * B * args.steps / elapsed
    roofline = None
    kernels = {}
    if timer is not None:
        # {(kernel, taps, CIN, N, H, W): (launches, total_ms, avg_ms, flops_per_launch, algorithmic_bytes_per_launch)}
        summ = timer.summary()
        tot_ms = sum(v[1] for v in summ.values())
        for k in sorted(summ, key=lambda k: -summ[k][1]):
            c, tms, ams, fl, nb = summ[k]
            log("  %-52s x%-3d %8.3f ms/step  avg %7.3f ms  %6.1f TFLOP/s %6.0f GB/s" % (
                "%s t%d CIN%d N%d %dx%d" % k, c // args.steps, tms / args.steps, ams, fl / ams / 1e9, nb / ams / 1e6))
        # the roofline object is quoted per KERNEL (template instance, the unit rocprofv3 --stats aggregates on):
        # all its launches of the timed region, algorithmic FLOPs = 2 * positions * taps * CIN * N of each launch
        byk = {}
        for k, (c, tms, ams, fl, nb) in summ.items():
            a = byk.setdefault(k[0], [0, 0.0, 0.0, 0.0])
            a[0] += c; a[1] += tms; a[2] += fl * c; a[3] += nb * c
        for name in sorted(byk, key=lambda n: -byk[n][1]):
            c, tms, fl, nb = byk[name]
            log("  KERNEL %-34s x%-3d %8.3f ms/step  avg %7.3f ms  %6.1f TFLOP/s %6.0f GB/s" % (
                name, c // args.steps, tms / args.steps, tms / c, fl / tms / 1e9, nb / tms / 1e6))
        name = max(byk, key=lambda n: byk[n][1])
        launches, total_ms, flops_total, bytes_total = byk[name]
        avg_ms = total_ms / launches
        tflops = flops_total / (total_ms * 1e-3) / 1e12
        gbs = bytes_total / (total_ms * 1e-3) / 1e9
        # kernels named *3_kernel / *3n_kernel run split-fp32 operands on the bf16 matrix cores: 3 bf16 MFMAs per
        # product, so their ceiling in algorithmic (fp32-equivalent) FLOP/s is the dense bf16 peak / 3
        split = re.search(r"3[a-z]?_kernel", name) is not None
        peak_tf = PEAK_BF16_MFMA_TFLOPS / 3.0 if split else PEAK_FP32_MFMA_TFLOPS
        # which roof bounds this kernel: its arithmetic intensity against the ridge point of ITS matrix-core ceiling
        intensity = flops_total / max(bytes_total, 1.0)
        ridge = peak_tf * 1e12 / (PEAK_HBM_GBS * 1e9)
        basis = ("dense bf16 MFMA 2500 / 3 (bf16x3 split-fp32 operands)" if split
                 else "fp32 MFMA 157.3 (v_mfma_f32_32x32x2_f32)")
        common = {"traffic": None, "kernel": name, "avg_launch_ms": round(avg_ms, 4), "launches": launches,
                  "algorithmic_gflop_per_launch": round(flops_total / launches / 1e9, 3),
                  "algorithmic_mbytes_per_launch": round(bytes_total / launches / 1e6, 2),
                  "intensity_flop_per_byte": round(intensity, 1), "ridge_flop_per_byte": round(ridge, 1),
                  "mfma_tflops": round(tflops, 2), "mfma_peak_tflops": round(peak_tf, 1), "mfma_peak_basis": basis,
                  "hbm_gbs": round(gbs, 1), "share_of_mfma_kernel_time": round(total_ms / tot_ms, 3),
                  "mfma_kernels_ms_per_step": round(tot_ms / args.steps, 3)}
        if intensity >= ridge:
            roofline = {"bound": "mfma", "achieved": round(tflops, 3), "peak": round(peak_tf, 1), "unit": "TFLOP/s",
                        "frac": round(tflops / peak_tf, 4)}
        else:
            roofline = {"bound": "hbm", "achieved": round(gbs, 1), "peak": PEAK_HBM_GBS, "unit": "GB/s",
                        "frac": round(gbs / PEAK_HBM_GBS, 4)}
        roofline.update(common)
        all_flops = sum(v[0] * v[3] for v in summ.values())
        kernels = {"all_mfma_kernels_tflops": round(all_flops / (tot_ms * 1e-3) / 1e12, 2)}
        # HBM bytes per launch of that kernel from the PMC passes (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE, separate
        # passes, FETCH doubled as MI355X_MICROARCH.md prescribes for gfx950); only valid for the profiled workload
        try:
            if args.mode == "crnn" and B == 256 and args.sr == 22050 and args.seconds == 10.0:
                pmc = json.load(open(os.path.join(ROOT, "profiles", "r01_pmc_traffic.json")))["kernels"].get(name)
                if pmc:
                    roofline["traffic"] = round(pmc["hbm_bytes_per_launch"])
                    roofline["traffic_source"] = "profiles/r01_pmc_traffic.json"
        except (OSError, KeyError, ValueError):
            pass
    cpu = None
    if not args.no_cpu_baseline:
        try:
            ncpu = len(os.sched_getaffinity(0))
        except AttributeError:
            ncpu = os.cpu_count() or 1
        cpu = cpu_baseline(args.sr, args.seconds, max(1, min(ncpu, 16)))
        log("cpu baseline done")
    workload = {
        "crnn": "waveform->STFT/mel/dB->CRNN(7 conv/BN/GLU/pool + 2xBiGRU128)->Predictor->BCE strong+weak"
                "->backward->Adam; BASELINE configs[2] (main_baseline.py train step on SYN)",
        "mt": "mean teacher: student on B/2 synthetic + B/2 real clips, EMA teacher forward on the noisy real half, "
              "consistency MSE, backward, Adam, EMA update; BASELINE configs[3] (main_scmt.py)",
        "ada": "domain-adversarial: student on B/2 synthetic + B/2 real clips, Clip_Discriminator + gradient reverse "
               "on both encodings, backward, Adam + SGD(discriminator); BASELINE configs[4] (main_scmt_ada_weak.py)",
    }[args.mode]
    line = {
        "metric": "10 s clips/sec through mel+CRNN train step", "value": round(value, 2), "unit": "clips/s",
        "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
        "ms_per_step": round(1e3 * elapsed / args.steps, 3), "higher_is_better": True, "scaling": "weak",
        "vs_baseline": None, "dtype": "f32", "data": "synthetic",
        "contraction_mode": os.environ.get("BSED_CONV_MODE", "bf16x3"),
        "config": {"workload": workload, "mode": args.mode,
                   "clip_seconds": args.seconds, "sr": args.sr, "frames": T, "out_frames": Tp,
                   "batch_per_gpu": B, "global_batch": B * world, "parallelism": f"dp{world}", "dropout": 0.5},
        "roofline": roofline, "cpu_baseline": cpu, "final_loss": round(loss, 5),
    }
    line.update(kernels)
    print(json.dumps(line))
    if torch.distributed.is_initialized():
        torch.distributed.destroy_process_group()


if __name__ == "__main__":
    main()
