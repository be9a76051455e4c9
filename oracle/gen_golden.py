#!/usr/bin/env python3
"""Generate tests/golden/* by RUNNING the reference's own Python (build container only).

The reference lives at /root/reference (read-only) and never travels to the GPU box;
only the vectors written here do.  Nothing in this script is copied from the
reference: modules are imported (with stub modules for uninstalled third-party
imports) and single functions of un-importable scripts are compiled from their own
file with ``ast`` at run time.

    python oracle/gen_golden.py            # writes tests/golden/
"""
import ast
import json
import os
import sys
import types

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
REF = "/root/reference/src"
OUT = os.environ.get("BSED_GOLDEN_OUT") or os.path.join(ROOT, "tests", "golden")   # tools/check_golden.py redirects
sys.path.insert(0, ROOT)
sys.path.insert(0, REF)

from oracle import seeded  # noqa: E402

np.float = float  # numpy-2 shim for src/DA/grl.py:64


def _stub(name, **attrs):
    m = types.ModuleType(name)
    m.__dict__.update(attrs)
    sys.modules[name] = m
    return m


# third-party modules the reference imports at top level but that are absent here
_stub("soundfile")
from oracle import mel_oracle as _mo  # noqa: E402
# librosa is absent: the reference's Transforms.py only calls librosa.amplitude_to_db, which is handed the
# restatement -- everything else that runs below (noise, pad/trunc, tensor conversion, pipeline order) is the reference's
_stub("librosa", amplitude_to_db=_mo.amplitude_to_db)
_stub("dcase_util")
_stub("dcase_util.data", DecisionEncoder=object)
# data.config allocates 2.6 GB at import: hand the constants in instead
_cfg = _stub("data.config", sr=32000, hop_size=255, pooling_time_ratio=4, max_learning_rate=0.0005,
             max_len_seconds=10.0)
import data  # noqa: E402  (namespace package of the reference)
data.config = _cfg

from models.CRNN_GRL import CRNN, Predictor, Clip_Discriminator  # noqa: E402
from DA.cdan_frame import ConditionalDomainAdversarialLoss  # noqa: E402
from utilities.utils import weights_init  # noqa: E402
from utilities import ramps  # noqa: E402
from utilities.ManyHotEncoder import ManyHotEncoder  # noqa: E402
from oracle.crnn_oracle import CRNN_KWARGS, PREDICTOR_KWARGS  # noqa: E402
from oracle.labels_oracle import BIRD_LIST  # noqa: E402


def ref_function(path, name, glb):
    """Compile ONE function of a reference script that cannot be imported as a module."""
    tree = ast.parse(open(path).read())
    fn = [n for n in tree.body if isinstance(n, ast.FunctionDef) and n.name == name]
    code = compile(ast.Module(body=fn, type_ignores=[]), path, "exec")
    exec(code, glb)
    return glb[name]


_glb = dict(torch=torch, np=np, cfg=_cfg)
update_ema_variables = ref_function(os.path.join(REF, "main_baseline.py"), "update_ema_variables", _glb)
adjust_learning_rate = ref_function(os.path.join(REF, "main_baseline.py"), "adjust_learning_rate", _glb)


def t(a):
    return torch.from_numpy(np.ascontiguousarray(a))


def build(dropout, seed):
    kw = dict(CRNN_KWARGS)
    kw["dropout"] = dropout
    crnn, pred = CRNN(**kw), Predictor(**PREDICTOR_KWARGS)
    v1 = seeded.load_seeded(crnn, seed)
    v2 = seeded.load_seeded(pred, seed + 1)
    return crnn, pred, seeded.checksum(v1), seeded.checksum(v2)


def named_grads(mods):
    out = {}
    for pfx, m in mods:
        for k, p in m.named_parameters():
            out[pfx + k] = p.grad.detach().numpy().copy()
    return out


def small_tensors(named, limit=4096):
    return {k: v for k, v in named.items() if v.size <= limit}


def crnn_case(tag, B, T, seed, adam_steps, mt):
    g = {}
    x = seeded.db_like_input(seed + 10, B, T)
    Tp = T // 4
    y = seeded.strong_targets(seed + 11, B, Tp)
    g["meta"] = np.array([B, T, seed], dtype=np.int64)
    # ---- eval mode (running stats, no dropout)
    crnn, pred, c1, c2 = build(0.5, seed)
    g["weight_checksum"] = np.array([c1, c2])
    crnn.eval(); pred.eval()
    with torch.no_grad():
        enc, _ = crnn(t(x))
        strong, weak = pred(enc)
    g["eval_enc"], g["eval_strong"], g["eval_weak"] = enc.numpy(), strong.numpy(), weak.numpy()
    # ---- train mode, dropout = 0 (batch statistics), syn-only loss + Adam
    crnn, pred, _, _ = build(0.0, seed)
    crnn.train(); pred.train()
    params = list(crnn.parameters()) + list(pred.parameters())
    opt = torch.optim.Adam(params, lr=1e-3, betas=(0.9, 0.999))
    bce = torch.nn.BCELoss()
    losses = []
    for step in range(adam_steps):
        opt.zero_grad()
        enc, _ = crnn(t(x))
        strong, weak = pred(enc)
        loss = bce(strong, t(y)) + bce(weak, t(y).max(-2)[0])
        loss.backward()
        if step == 0:
            g["train_enc"], g["train_strong"], g["train_weak"] = (
                enc.detach().numpy(), strong.detach().numpy(), weak.detach().numpy())
            grads = named_grads([("crnn.", crnn), ("pred.", pred)])
            g["grad_names"] = np.array(list(grads.keys()))
            g["grad_norms"] = np.array([np.sqrt((v.astype(np.float64) ** 2).sum()) for v in grads.values()])
            for k, v in small_tensors(grads).items():
                g["grad/" + k] = v
        opt.step()
        losses.append(float(loss))
        if step in (0, adam_steps - 1):
            sd = {"crnn." + k: v.numpy() for k, v in crnn.state_dict().items()}
            sd.update({"pred." + k: v.numpy() for k, v in pred.state_dict().items()})
            g[f"adam{step+1}_names"] = np.array(list(sd.keys()))
            g[f"adam{step+1}_norms"] = np.array([np.sqrt((v.astype(np.float64) ** 2).sum()) for v in sd.values()])
            for k, v in small_tensors(sd, 512).items():
                g[f"adam{step+1}/" + k] = v
    g["train_losses"] = np.array(losses)
    if mt:
        # ---- mean-teacher iteration: student(real) vs EMA(noisy real), src/main_baseline.py:337-368,431-498
        crnn, pred, _, _ = build(0.0, seed)
        ema_c, ema_p, _, _ = build(0.0, seed + 5)
        for m in (crnn, pred, ema_c, ema_p):
            m.train()
        for p in list(ema_c.parameters()) + list(ema_p.parameters()):
            p.detach_()
        xr = seeded.db_like_input(seed + 20, B, T)
        xe = xr + np.random.default_rng(seed + 21).normal(0, 1.0, xr.shape).astype(np.float32)
        yw = (np.random.default_rng(seed + 22).random((B, 20)) < 0.2).astype(np.float32)
        mse = torch.nn.MSELoss()
        opt = torch.optim.Adam(list(crnn.parameters()) + list(pred.parameters()), lr=1e-3, betas=(0.9, 0.999))
        opt.zero_grad()
        enc_s, _ = crnn(t(x)); ss, ws = pred(enc_s)
        enc_r, _ = crnn(t(xr)); sr_, wr = pred(enc_r)
        enc_e, _ = ema_c(t(xe)); se, we = ema_p(enc_e)
        se, we = se.detach(), we.detach()
        w = 0.7
        loss = (bce(ss, t(y)) + bce(ws, t(y).max(-2)[0]) + bce(wr, t(yw))
                + w * mse(sr_, se) + w * mse(wr, we))
        loss.backward()
        opt.step()
        g["mt_loss"] = np.array(float(loss))
        g["mt_strong_ema"], g["mt_weak_ema"] = se.numpy(), we.numpy()
        g["mt_strong_real"], g["mt_weak_real"] = sr_.detach().numpy(), wr.detach().numpy()
        grads = named_grads([("crnn.", crnn), ("pred.", pred)])
        g["mt_grad_norms"] = np.array([np.sqrt((v.astype(np.float64) ** 2).sum()) for v in grads.values()])
        for gs in (1, 5000):
            # The reference call update_ema_variables(crnn, ema_crnn, ...) RAISES for a plain CRNN:
            # CNN.state_dict() drops a "cnn." level that load_state_dict() then misses.
            # Its arithmetic is pinned by running it on the consistent sub-modules instead.
            update_ema_variables(crnn.cnn.cnn, ema_c.cnn.cnn, 0.999, gs)
            update_ema_variables(crnn.rnn, ema_c.rnn, 0.999, gs)
            update_ema_variables(pred, ema_p, 0.999, gs)
            sd = {"crnn." + k: v.numpy().copy() for k, v in ema_c.state_dict().items()}
            sd.update({"pred." + k: v.numpy().copy() for k, v in ema_p.state_dict().items()})
            g[f"ema{gs}_names"] = np.array(list(sd.keys()))
            g[f"ema{gs}_norms"] = np.array([np.sqrt((v.astype(np.float64) ** 2).sum()) for v in sd.values()])
            for k, v in small_tensors(sd, 512).items():
                g[f"ema{gs}/" + k] = v
    np.savez_compressed(os.path.join(OUT, f"crnn_{tag}.npz"), **g)
    print("wrote crnn_%s.npz" % tag, {k: getattr(v, "shape", None) for k, v in list(g.items())[:6]})


def clipd_case():
    seed = 77
    B, T = 2, 313
    rng = np.random.default_rng(seed)
    f_s = rng.standard_normal((B, T, 256)).astype(np.float32)
    f_t = rng.standard_normal((B, T, 256)).astype(np.float32)
    g_s = rng.random((B, T, 20)).astype(np.float32)
    g_t = rng.random((B, T, 20)).astype(np.float32)
    disc = Clip_Discriminator(input_dim=8192, dropout=0.5)
    vals = seeded.load_seeded(disc, seed + 1)
    disc.train()
    cdan = ConditionalDomainAdversarialLoss(disc, entropy_conditioning=False, num_classes=20,
                                            features_dim=256, randomized=False)
    g = {"meta": np.array([B, T, seed]), "weight_checksum": np.array(seeded.checksum(vals))}
    for it in range(3):
        fs, ft = t(f_s).requires_grad_(), t(f_t).requires_grad_()
        disc.zero_grad()
        loss = cdan(t(g_s), fs, t(g_t), ft)
        loss.backward()
        g[f"loss{it}"] = np.array(float(loss))
        g[f"dfs_norm{it}"] = np.array(float(fs.grad.norm()))
        g[f"dft_norm{it}"] = np.array(float(ft.grad.norm()))
        g[f"dfs{it}"] = fs.grad.numpy()[:, ::16, ::8].copy()
        gr = named_grads([("", disc)])
        g[f"dnames"] = np.array(list(gr.keys()))
        g[f"dgrad_norms{it}"] = np.array([np.sqrt((v.astype(np.float64) ** 2).sum()) for v in gr.values()])
    with torch.no_grad():
        disc.eval()
        g["eval_out"] = disc(t(np.concatenate([f_s, f_t]))).numpy()
    np.savez_compressed(os.path.join(OUT, "clipd.npz"), **g)
    print("wrote clipd.npz")


def init_case():
    torch.manual_seed(1)            # construction draws (the GRU biases keep them: weights_init skips 1-D GRU tensors)
    crnn, pred = CRNN(**CRNN_KWARGS), Predictor(**PREDICTOR_KWARGS)
    torch.manual_seed(2023)         # re-seeded AFTER construction: the draws below are weights_init's alone
    crnn.apply(weights_init)
    pred.apply(weights_init)
    stats = {}
    for pfx, m in (("crnn.", crnn), ("pred.", pred)):
        for k, v in m.state_dict().items():
            v = v.double()
            stats[pfx + k] = [float(v.mean()), float(v.std()) if v.numel() > 1 else 0.0,
                              float(v.abs().max()), float(v.abs().sum())]
    w = crnn.state_dict()["rnn.rnn.weight_hh_l0"].double()
    stats["_gru_hh_gram_err"] = float((w.T @ w - torch.eye(128, dtype=torch.double)).abs().max())
    json.dump(stats, open(os.path.join(OUT, "weights_init.json"), "w"), indent=0)
    print("wrote weights_init.json")


def schedule_case():
    out = {"sigmoid_rampdown_30": [ramps.sigmoid_rampdown(e, 30) for e in range(0, 40, 3)],
           "exp_rampup_50": [ramps.exp_rampup(e, 50) for e in range(0, 60, 5)]}

    class _Opt:
        def __init__(self):
            self.param_groups = [{"lr": 0.0}]
    lrs = []
    for e in (0, 10, 29, 30, 99, 100, 101, 120, 121, 141, 299):
        o, od, oc = _Opt(), _Opt(), _Opt()
        adjust_learning_rate(o, ramps.sigmoid_rampdown(e, 30), optimizer_d=od, optimizer_crnn=oc, c_epoch=e)
        lrs.append([e, o.param_groups[0]["lr"], od.param_groups[0]["lr"], oc.param_groups[0]["lr"]])
    out["adjust_learning_rate"] = lrs
    from DA.grl import WarmStartGradientReverseLayer
    grl = WarmStartGradientReverseLayer(alpha=1., lo=0., hi=1., max_iters=1000, auto_step=True)
    coeffs = []
    for it in range(5):
        x = torch.ones(1, requires_grad=True)
        grl(x).sum().backward()
        coeffs.append(float(-x.grad))
    out["grl_coeff_first5"] = coeffs
    json.dump(out, open(os.path.join(OUT, "schedules.json"), "w"), indent=0)
    print("wrote schedules.json")


def labels_case():
    import pandas as pd
    df = pd.read_csv("/root/reference/dataset/SYN/generated/output.tsv", sep="\t")
    enc = ManyHotEncoder(BIRD_LIST, n_frames=313)
    cases = []
    for fname in list(dict.fromkeys(df.filename))[:12]:
        sub = df[df.filename == fname]
        y = enc.encode_strong_df(sub)
        rows = [[float(r.onset), float(r.offset), r.event_label] for r in sub.itertuples()]
        nz = np.argwhere(y > 0)
        cases.append({"filename": fname, "events": rows, "sum": float(y.sum()),
                      "first_last": {c: [int(nz[nz[:, 1] == BIRD_LIST.index(c), 0].min()),
                                         int(nz[nz[:, 1] == BIRD_LIST.index(c), 0].max()) + 1]
                                     for c in sorted(set(sub.event_label))
                                     if (nz[:, 1] == BIRD_LIST.index(c)).any()},
                      "weak": enc.encode_weak(list(sub.event_label)).tolist(),
                      "col_sums": y.sum(0).tolist()})
    json.dump(cases, open(os.path.join(OUT, "labels_kat.json"), "w"), indent=0)
    print("wrote labels_kat.json")


def fpn_case():
    """CRNN_fpn of the reference (CRNN_GRL.py:293-389) at the only length its hard-coded Upsample sizes admit:
    1255 input frames -> 313 / 156 / 78.  Train-mode vectors are taken with BOTH dropouts off (the FPN levels carry a
    fixed Dropout(0.5): its p is set to 0 on the instance) so that gradients are comparable."""
    from models.CRNN_GRL import CRNN_fpn as RefFPN
    B, T, seed = 2, 1255, 31
    g = {"meta": np.array([B, T, seed], dtype=np.int64)}
    x = seeded.db_like_input(seed + 10, B, T)
    kw = dict(CRNN_KWARGS)
    kw["dropout"] = 0.5
    m = RefFPN(**kw)
    vals = seeded.load_seeded(m, seed)
    g["weight_checksum"] = np.array([seeded.checksum(vals)])
    g["state_names"] = np.array(list(m.state_dict().keys()))
    m.eval()
    with torch.no_grad():
        enc, d_in = m(t(x))
    g["eval_enc"] = enc.numpy()
    kw["dropout"] = 0.0
    m = RefFPN(**kw)
    seeded.load_seeded(m, seed)
    m.cnn.dropout.p = 0.0
    m.train()
    enc, _ = m(t(x))
    w = torch.sin(torch.arange(enc.numel(), dtype=torch.float32)).view_as(enc) * 1e-2
    (enc * w).sum().backward()
    g["train_enc"] = enc.detach().numpy()
    grads = {k: p.grad.detach().numpy().copy() for k, p in m.named_parameters() if p.grad is not None}
    g["grad_names"] = np.array(list(grads.keys()))
    g["grad_norms"] = np.array([float(np.linalg.norm(v.astype(np.float64))) for v in grads.values()])
    for k, v in small_tensors(grads).items():
        g["grad/" + k] = v
    sd = m.state_dict()
    for k in ("cnn.bn_fcn.running_mean", "cnn.bn_fcn.running_var", "cnn.bn_fcn.num_batches_tracked"):
        g["after/" + k] = sd[k].numpy().copy()
    np.savez_compressed(os.path.join(OUT, "crnn_fpn.npz"), **g)
    print("wrote crnn_fpn.npz", {k: (v.shape if hasattr(v, "shape") else v) for k, v in list(g.items())[:6]})


def transforms_case():
    """AugmentGaussianNoise.gaussian_noise / pad_trunc_seq / PadOrTrunc / ToTensor and the get_transforms pipeline
    of the reference (src/data/Transforms.py:89-139,155-227,304-322), run on seeded linear-mel inputs.  Only
    librosa.amplitude_to_db inside ApplyLog is the restatement's (librosa is not installed)."""
    from data import Transforms as TR
    rng = np.random.default_rng(404)
    T, F = 37, 128
    x = (np.abs(rng.standard_normal((T, F))) * np.exp(rng.uniform(-6, 2, (1, F)))).astype(np.float32)
    g = {"x": x}
    np.random.seed(2023)
    g["noisy_snr30"] = TR.AugmentGaussianNoise.gaussian_noise(x, 30)
    np.random.seed(7)
    x3 = np.stack([x, 0.5 * x])                         # the 3-D branch: std from features[0] only
    g["noisy3d_snr20"] = TR.AugmentGaussianNoise.gaussian_noise(x3, 20)
    g["pad50"] = TR.pad_trunc_seq(x, 50)
    g["trunc20"] = TR.pad_trunc_seq(x, 20)
    g["same37"] = TR.pad_trunc_seq(x, 37)
    g["pad3d_40"] = TR.pad_trunc_seq(x3, 40)
    lab = (rng.random((9, 20)) < 0.3).astype(np.float64)
    for tag, frames in (("pad", 50), ("trunc", 20)):
        np.random.seed(99)
        tf = TR.get_transforms(frames, None, 0, noise_dict_params={"mean": 0., "snr": 30})
        (clean, noisy), y = tf((x, lab))
        assert clean.dtype == torch.float32 and noisy.dtype == torch.float32 and y.dtype == torch.float32
        g[f"pipe_{tag}_clean"], g[f"pipe_{tag}_noisy"], g[f"pipe_{tag}_label"] = clean.numpy(), noisy.numpy(), y.numpy()
    np.savez_compressed(os.path.join(OUT, "transforms.npz"), **g)
    print("wrote transforms.npz", {k: v.shape for k, v in g.items()})


def fpn_init_case():
    """per-tensor statistics of the reference's CRNN_fpn after .apply(weights_init) (utilities/utils.py:40-63)"""
    from models.CRNN_GRL import CRNN_fpn as RefFPN
    torch.manual_seed(1)            # construction draws (GRU biases), see init_case
    m = RefFPN(**CRNN_KWARGS)
    torch.manual_seed(2023)         # re-seeded AFTER construction: the draws below are weights_init's alone
    m.apply(weights_init)
    stats = {}
    for k, v in m.state_dict().items():
        v = v.double()
        stats[k] = [float(v.mean()), float(v.std()) if v.numel() > 1 else 0.0, float(v.abs().max()),
                    float(v.abs().sum())]
    json.dump(stats, open(os.path.join(OUT, "weights_init_fpn.json"), "w"), indent=0)
    print("wrote weights_init_fpn.json", len(stats))


def frame_d_case():
    """the reference's Frame_Discriminator (models/CRNN_GRL.py:116-140): forward and, for a fixed upstream gradient, the
    gradients of every parameter and of the input (the reference defines no runnable loss for it)"""
    from models.CRNN_GRL import Frame_Discriminator
    seed, N, T = 83, 3, 157
    rng = np.random.default_rng(seed)
    x = rng.standard_normal((N, T, 256)).astype(np.float32)
    m = Frame_Discriminator(input_dim=256, dropout=0)
    vals = seeded.load_seeded(m, seed + 1)
    m.train()
    xt = t(x).requires_grad_()
    d = m(xt)
    up = (torch.cos(torch.arange(d.numel(), dtype=torch.float32)).view_as(d) * 0.3)
    (d * up).sum().backward()
    g = {"meta": np.array([N, T, seed]), "weight_checksum": np.array([seeded.checksum(vals)]), "out": d.detach().numpy(),
         "dx": xt.grad.numpy()[:, ::7, ::5].copy(), "dx_norm": np.array(float(xt.grad.norm())),
         "state_names": np.array(list(m.state_dict().keys()))}
    for k, p in m.named_parameters():
        g["grad/" + k] = p.grad.numpy().copy()
    np.savez_compressed(os.path.join(OUT, "frame_d.npz"), **g)
    print("wrote frame_d.npz", {k: getattr(v, "shape", None) for k, v in g.items()})


def cnn_pred_case():
    """BASELINE configs[1] (CNN-only tagging forward): the reference's CRNN_pred (models/CRNN_GRL.py:206-290) =
    CNN stack -> sigmoid on the 128 channel features, class-softmax attention pooling with dense_softmax; its GRU
    exists but forward never calls it.  Shapes only agree for nclass == nb_filters[-1] == 2*n_RNN_cell."""
    from models.CRNN_GRL import CRNN_pred
    kw = dict(CRNN_KWARGS)
    kw.update(nclass=128, n_RNN_cell=64)
    g = {}
    for tag, B, T, seed in (("small", 2, 64, 51), ("R", 2, 1255, 52)):
        x = seeded.db_like_input(seed + 10, B, T)
        kw["dropout"] = 0.5
        m = CRNN_pred(**kw)
        vals = seeded.load_seeded(m, seed)
        g[f"{tag}_meta"] = np.array([B, T, seed], dtype=np.int64)
        g[f"{tag}_weight_checksum"] = np.array([seeded.checksum(vals)])
        g["state_names"] = np.array(list(m.state_dict().keys()))
        g["state_shapes"] = np.array([str(tuple(v.shape)) for v in m.state_dict().values()])
        m.eval()
        with torch.no_grad():
            strong, weak = m(t(x))
        g[f"{tag}_eval_strong"], g[f"{tag}_eval_weak"] = strong.numpy(), weak.numpy()
        kw["dropout"] = 0.0
        m = CRNN_pred(**kw)
        seeded.load_seeded(m, seed)
        m.train()
        with torch.no_grad():
            strong, weak = m(t(x))
        g[f"{tag}_train_strong"], g[f"{tag}_train_weak"] = strong.numpy(), weak.numpy()
        sd = m.state_dict()
        g[f"{tag}_after_rm6"] = sd["cnn.batchnorm6.running_mean"].numpy().copy()
        g[f"{tag}_after_rv6"] = sd["cnn.batchnorm6.running_var"].numpy().copy()
    np.savez_compressed(os.path.join(OUT, "cnn_pred.npz"), **g)
    print("wrote cnn_pred.npz", {k: getattr(v, "shape", None) for k, v in g.items()})


def isp_case():
    """One ``-mt -ISP`` iteration of train_mt, re-assembled from the IMPORTED reference CRNN / Predictor with the
    reference's own per-sample torch.roll loops and loss composition (src/main_baseline.py:229-277 batch views,
    :337-420 forwards and rolled predictions / targets, :442-529 loss).  Written against the script, not against
    oracle/crnn_oracle.train_losses_isp, which this fixture pins (tests/test_oracle_golden.py) together with the HIP
    step (tests/test_crnn_gpu.py).  Same seeds / sizes as that GPU test: B = 4 + 4, T = 128."""
    seed, B, T = 41, 4, 128
    rng = np.random.default_rng(seed)
    xs = seeded.db_like_input(seed + 1, B, T); xr = seeded.db_like_input(seed + 2, B, T)
    xe = xr + rng.normal(0, 1.0, xr.shape).astype(np.float32)
    y = seeded.strong_targets(seed + 3, B, T // 4)
    yw = (rng.random((B, 20)) < 0.2).astype(np.float32)
    shift_list, freq_shift_list = [-8, 12, 0, 40], [3, -2, 0, -4]     # randint(-64,64)*4 / randint(-4,4) draws
    cc = 0.6                                                          # cfg.max_consistency_cost * rampup_value
    model, predictor, c1, c2 = build(0.0, seed)
    ema_model, ema_predictor, c3, c4 = build(0.0, seed + 5)
    for m in (model, predictor, ema_model, ema_predictor):
        m.train()
    for p in list(ema_model.parameters()) + list(ema_predictor.parameters()):
        p.detach_()
    class_criterion, consistency_criterion = torch.nn.BCELoss(), torch.nn.MSELoss()
    syn_batch_input, syn_target = t(xs), t(y)
    batch_input, ema_batch_input, target_weak = t(xr), t(xe), t(yw)
    pooling_time_ratio = 4
    views = {}
    for name, src in (("batch", batch_input), ("ema", ema_batch_input), ("syn", syn_batch_input)):
        sh, fs = [], []
        for k in range(batch_input.shape[0]):                          # :233-246 (sample k is (1, T, F))
            sh.append(torch.unsqueeze(torch.roll(src[k], shift_list[k], dims=1), 0))
            fs.append(torch.unsqueeze(torch.roll(src[k], freq_shift_list[k], dims=2), 0))
        views[name + "_shift"], views[name + "_freq_shift"] = torch.cat(sh, 0), torch.cat(fs, 0)
    syn_encoded_x, _ = model(syn_batch_input)                          # :337-341
    syn_strong_pred, syn_weak_pred = predictor(syn_encoded_x)
    encoded_x, _ = model(batch_input)
    strong_pred, weak_pred = predictor(encoded_x)
    strong_pred_ema, weak_pred_ema = [v.detach() for v in ema_predictor(ema_model(ema_batch_input)[0])]   # :352-356
    strong_pred_shift_ema = ema_predictor(ema_model(views["ema_shift"])[0])[0].detach()                   # :360-363
    strong_pred_freq_shift_ema = ema_predictor(ema_model(views["ema_freq_shift"])[0])[0].detach()         # :365-368
    sp, ssp, sts = [], [], []
    for k in range(strong_pred.shape[0]):                              # :374-401
        pool_shift = int(shift_list[k] / pooling_time_ratio)
        sp.append(torch.unsqueeze(torch.roll(strong_pred[k], pool_shift, dims=0), 0))
        ssp.append(torch.unsqueeze(torch.roll(syn_strong_pred[k], pool_shift, dims=0), 0))
        sts.append(torch.unsqueeze(torch.roll(syn_target[k], pool_shift, dims=0), 0))
    strong_pred_shift, syn_strong_pred_shift = torch.cat(sp, 0).detach(), torch.cat(ssp, 0).detach()
    syn_strong_target_shift = torch.cat(sts, 0)
    strong_shift_pred, weak_shift_pred = predictor(model(views["batch_shift"])[0])                        # :408-419
    strong_freq_shift_pred, weak_freq_shift_pred = predictor(model(views["batch_freq_shift"])[0])
    syn_strong_shift_pred, _ = predictor(model(views["syn_shift"])[0])
    syn_strong_freq_shift_pred, syn_weak_freq_shift_pred = predictor(model(views["syn_freq_shift"])[0])
    syn_target_weak = syn_target.max(-2)[0]                            # :431-447
    weak_class_loss = class_criterion(syn_weak_pred, syn_target_weak)
    weak_index = target_weak.shape[0] // 2
    weak_class_loss = weak_class_loss + class_criterion(weak_pred, target_weak)
    weak_freq_shift_class_loss = (class_criterion(syn_weak_freq_shift_pred, syn_target_weak)
                                  + class_criterion(weak_freq_shift_pred[:weak_index], target_weak[:weak_index]))
    strong_class_loss = class_criterion(syn_strong_pred, syn_target)   # :477-483
    strong_shift_class_loss = class_criterion(syn_strong_shift_pred, syn_strong_target_shift)
    strong_freq_shift_class_loss = class_criterion(syn_strong_freq_shift_pred, syn_target)
    consistency_loss_strong = cc * consistency_criterion(strong_pred, strong_pred_ema)                    # :486-497
    consistency_loss_weak = cc * consistency_criterion(weak_pred, weak_pred_ema)
    consistency_loss_strong_shift = cc * consistency_criterion(strong_shift_pred, strong_pred_shift_ema)  # :499-512
    consistency_loss_strong_freq_shift = cc * consistency_criterion(strong_freq_shift_pred, strong_pred_freq_shift_ema)
    loss = strong_class_loss + weak_class_loss                         # :516-529
    loss = loss + (consistency_loss_weak + consistency_loss_strong)
    consistency_loss_shift = cc / 2 * (consistency_criterion(syn_strong_shift_pred, syn_strong_pred_shift)
                                       + consistency_criterion(strong_shift_pred, strong_pred_shift))
    loss = loss + (weak_freq_shift_class_loss + strong_shift_class_loss + strong_freq_shift_class_loss
                   + consistency_loss_shift)
    loss = loss + 1 / 2 * (consistency_loss_strong_shift + consistency_loss_strong_freq_shift)
    loss.backward()
    g = {"meta": np.array([B, T, seed], dtype=np.int64), "weight_checksum": np.array([c1, c2, c3, c4]),
         "shift_frames": np.array(shift_list), "shift_bins": np.array(freq_shift_list),
         "consistency_cost": np.array(cc), "loss": np.array(float(loss)),
         "parts": np.array([float(v) for v in (strong_class_loss, weak_class_loss, consistency_loss_weak,
                                               consistency_loss_strong, weak_freq_shift_class_loss,
                                               strong_shift_class_loss, strong_freq_shift_class_loss,
                                               consistency_loss_shift, consistency_loss_strong_shift,
                                               consistency_loss_strong_freq_shift)]),
         "strong_shift_pred": strong_shift_pred.detach().numpy(),
         "syn_strong_freq_shift_pred": syn_strong_freq_shift_pred.detach().numpy()}
    grads = named_grads([("crnn.", model), ("pred.", predictor)])
    g["grad_names"] = np.array(list(grads.keys()))
    g["grad_norms"] = np.array([np.sqrt((v.astype(np.float64) ** 2).sum()) for v in grads.values()])
    for k, v in small_tensors(grads).items():
        g["grad/" + k] = v
    np.savez_compressed(os.path.join(OUT, "isp.npz"), **g)
    print("wrote isp.npz loss", float(loss))


if __name__ == "__main__":
    os.makedirs(OUT, exist_ok=True)
    torch.set_num_threads(8)
    only = sys.argv[1:]
    if only:                       # python oracle/gen_golden.py transforms_case fpn_init_case ...
        for name in only:
            globals()[name]()
        sys.exit(0)
    labels_case()
    schedule_case()
    init_case()
    clipd_case()
    crnn_case("small", B=2, T=64, seed=11, adam_steps=3, mt=True)
    crnn_case("R", B=2, T=1255, seed=23, adam_steps=1, mt=False)
    fpn_case()
    transforms_case()
    fpn_init_case()
    cnn_pred_case()
    frame_d_case()
    isp_case()
