"""Experiment: the next batch's mel transform on a side stream beside the (half-chip, latency-bound) GRU forward."""
import os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
from bsed_amd.engine import FlatAdam, SEDTrainer
from bsed_amd.features import MelConfig, MelFrontEnd
from bsed_amd.models import CRNN, Predictor, weights_init

dev = torch.device("cuda:0")
kw = dict(n_in_channel=1, nclass=20, attention=True, n_RNN_cell=128, n_layers_RNN=2, activation="glu", dropout=0.5,
          kernel_size=7 * [3], padding=7 * [1], stride=7 * [1], nb_filters=[16, 32, 64, 128, 128, 128, 128],
          pooling=[[2, 2], [2, 2], [1, 2], [1, 2], [1, 2], [1, 2], [1, 2]])
torch.manual_seed(2023)
mcfg = MelConfig(sr=22050)
fe = MelFrontEnd(mcfg)
B, n = 256, 220500
wav, ev = bench.synth_waves(B, n, 22050, 2023, dev)
T = fe.num_frames(n)
crnn, pred = CRNN(**kw), Predictor(nclass=20, attention=True, n_RNN_cell=128)
weights_init(crnn); weights_init(pred)
tr = SEDTrainer(crnn, pred, optimizer=FlatAdam([crnn, pred], lr=1e-3), frontend=fe, seed=2023)
y = bench.strong_labels(ev, T // 4, 22050, mcfg.hop_size, 4, dev)
inp = fe.transform(wav, max_frames=T)
side = torch.cuda.Stream()
keep = []


def hook():
    side.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(side):
        keep[:] = [fe.transform(wav, max_frames=T)]


def run(name, step, K=30):
    for _ in range(3):
        step()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(K):
        step()
    torch.cuda.synchronize()
    print(f"{name}: {(time.perf_counter() - t0) / K * 1e3:.3f} ms/step", flush=True)


def step_b():
    crnn.rnn_hook = hook     # one-shot in the model: re-armed for every step
    out = tr.train_step(inp, y)
    torch.cuda.current_stream().wait_stream(side)
    return out


run("A mel inside the step", lambda: tr.train_step(wav, y, from_wave=True))
run("B mel on the side stream beside the GRU forward", step_b)
crnn.rnn_hook = None
run("C no mel at all (bound)", lambda: tr.train_step(inp, y))
run("A again", lambda: tr.train_step(wav, y, from_wave=True))

# D: the trainer's own two-deep pipeline (train_step(next_waves=...)) over two alternating batches, as bench.py runs it
wav2, ev2 = bench.synth_waves(B, n, 22050, 4046, dev)
y2 = bench.strong_labels(ev2, T // 4, 22050, mcfg.hop_size, 4, dev)
batches = [(wav, y), (wav2, y2)]
cnt = [0]


def step_d():
    (w0, y0), (w1, _) = batches[cnt[0] % 2], batches[(cnt[0] + 1) % 2]
    cnt[0] += 1
    return tr.train_step(w0, y0, from_wave=True, next_waves=(w1, None))


run("D trainer pipeline, two alternating batches", step_d)
run("C again", lambda: tr.train_step(inp, y))
run("D again", step_d)
