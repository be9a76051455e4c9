"""On-disk formats, checkpoint dictionary and the GPU event post-processing kernel."""
import os

import numpy as np
import pytest
import scipy.ndimage
import torch

from oracle import crnn_oracle as co
from oracle import labels_oracle as lo
from oracle import mel_oracle as mo
from oracle import seeded

pytestmark = pytest.mark.gpu


def test_binarize_median_matches_scipy():
    from bsed_amd.evaluation import binarize_median_gpu
    rng = np.random.default_rng(0)
    for T, win in ((313, 14), (216, 9), (40, 1), (17, 14), (5, 7)):
        p = rng.random((3, T, 20)).astype(np.float32)
        # long runs so that the filter has something to keep
        p[:, T // 4: T // 2, :5] += 0.6
        got = binarize_median_gpu(torch.from_numpy(p).cuda(), 0.5, win).cpu().numpy()
        for b in range(3):
            ref = scipy.ndimage.median_filter((p[b] > 0.5).astype(np.float64), (win, 1))
            np.testing.assert_array_equal(got[b], ref.astype(np.float32), err_msg=f"T={T} win={win}")


def test_feature_files_roundtrip_and_gpu_collate(tmp_path):
    from bsed_amd.data import FeatureDataset, GpuCollate, write_features
    from bsed_amd.features import MelFrontEnd
    from bsed_amd.labels import BIRD_LIST, ManyHotEncoder
    fe = MelFrontEnd()
    enc = ManyHotEncoder(BIRD_LIST, n_frames=313)
    root = str(tmp_path / "prep")
    lengths = [2.0, 2.0, 1.5]   # ragged: the last clip is shorter (final segment of a recording)
    for i, sec in enumerate(lengths):
        y, ev = mo.synth_clip(i, seconds=sec)
        write_features(root, f"clip{i}", y, [(on, off, BIRD_LIST[c]) for on, off, c in ev], frontend=fe)
    ds = FeatureDataset(root, enc.encode_strong_df)
    assert len(ds) == 3
    (feat0, tgt0), path0 = ds[0]
    assert feat0.dtype == np.float32 and feat0.shape == (1 + 64000 // 255, 128) and tgt0.shape == (313, 20)
    max_frames = 1 + 64000 // 255 + 5
    collate = GpuCollate(fe, max_frames=max_frames, noisy=True, seed=3)
    ((x, xn), target), paths = collate([ds[i] for i in range(3)])
    assert x.shape == (3, 1, max_frames, 128) and xn.shape == x.shape and target.shape == (3, 313, 20)
    for i in range(3):
        (feat, _), _ = ds[i]
        ref, _ = mo.transform_pair(feat, max_frames, unit_noise=np.zeros(feat.shape))
        live = ref[0, :feat.shape[0]] > ref[0, :feat.shape[0]].max() - 80 + 1e-3
        assert np.abs(x[i, 0, :feat.shape[0]].cpu().numpy() - ref[0, :feat.shape[0]])[live].max() < 2e-3
        assert float(x[i, 0, feat.shape[0]:].abs().max()) == 0.0     # zero padding in the dB domain
    assert float((x - xn).abs().max()) > 0.0                          # the EMA view really is noisy


def test_checkpoint_dictionary_keys_and_reload(tmp_path):
    from bsed_amd import checkpoint
    from bsed_amd.engine import FlatAdam
    from bsed_amd.labels import BIRD_LIST, ManyHotEncoder
    from bsed_amd.models import CRNN, Predictor
    crnn, pred = CRNN(**co.CRNN_KWARGS), Predictor(**co.PREDICTOR_KWARGS)
    ema_c, ema_p = CRNN(**co.CRNN_KWARGS), Predictor(**co.PREDICTOR_KWARGS)
    st = checkpoint.build_state(crnn, pred, co.CRNN_KWARGS, co.PREDICTOR_KWARGS, optimizer=FlatAdam([crnn, pred]),
                                many_hot_encoder=ManyHotEncoder(BIRD_LIST, 313), epoch=7, crnn_ema=ema_c,
                                predictor_ema=ema_p)
    for k in ("model", "model_p", "model_ema", "model_p_ema", "optimizer", "pooling_time_ratio", "many_hot_encoder",
              "median_window", "epoch"):
        assert k in st
    assert set(st["model"].keys()) == {"name", "args", "kwargs", "state_dict"} and st["model"]["name"] == "CRNN"
    # the stored CRNN state dict loads into the reference-architecture oracle (after its loaders' key rewrite) ...
    ocrnn = co.CRNN(**co.CRNN_KWARGS)
    ocrnn.load_state_dict({("cnn." + k if k.startswith("cnn.") else k): v for k, v in st["model"]["state_dict"].items()})
    # ... and round-trips through a file
    path = str(tmp_path / "baseline_epoch_7")
    checkpoint.save(st, path)
    back = checkpoint.load_models(path)
    assert torch.equal(back["crnn"].flat, crnn.flat) and torch.equal(back["predictor_ema"].flat, ema_p.flat)
    assert back["state"]["epoch"] == 7


def test_pseudo_weak_dataset_matches_reference_semantics(tmp_path):
    """ENA_Dataset_unlabeled (reference src/data/dataload.py:84-126): weak targets of an unlabeled clip = encode_weak of
    the event_labels of the TSV rows whose filename equals the feature path; no row -> all zeros"""
    import pandas as pd
    from bsed_amd.data import ENA_Dataset_unlabeled, PseudoWeakDataset
    from bsed_amd.labels import BIRD_LIST, ManyHotEncoder
    assert ENA_Dataset_unlabeled is PseudoWeakDataset
    root = tmp_path / "unl"
    (root / "wav").mkdir(parents=True)
    rng = np.random.default_rng(1)
    for i in range(3):
        np.save(root / "wav" / f"Recording_1_Segment_0{i}_4.npy", rng.random((40, 128)).astype(np.float32))
    paths = sorted(str(p) for p in (root / "wav").glob("*.npy"))
    tsv = tmp_path / "pseudo.tsv"
    pd.DataFrame({"filename": [paths[0], paths[2], "/elsewhere/wav/x.npy"],
                  "event_labels": ["EATO,BLJA", "NOCA", "AMCR"]}).to_csv(tsv, sep="\t", index=False)
    enc = ManyHotEncoder(BIRD_LIST, n_frames=313)
    ds = PseudoWeakDataset(str(root), enc.encode_weak, pseudo_label_tsv=str(tsv))
    assert len(ds) == 3
    want = [{"EATO", "BLJA"}, set(), {"NOCA"}]
    for i in range(3):
        (feat, target), path = ds[i]
        assert path == paths[i] and feat.shape == (40, 128) and target.shape == (20,)
        assert {BIRD_LIST[k] for k in np.nonzero(target)[0]} == want[i]
    # a relocated data set: match by file name
    moved = tmp_path / "pseudo_moved.tsv"
    pd.DataFrame({"filename": ["/home/other/wav/" + os.path.basename(paths[1])], "event_labels": ["BAWW,EATO"]}).to_csv(
        moved, sep="\t", index=False)
    ds2 = PseudoWeakDataset(str(root), enc.encode_weak, pseudo_label_tsv=str(moved), match="basename")
    (_, t1), _ = ds2[1]
    assert {BIRD_LIST[k] for k in np.nonzero(t1)[0]} == {"BAWW", "EATO"} and ds2[0][0][1].sum() == 0


def test_optimizer_state_dicts_are_torch_compatible():
    """FlatAdam / FlatSGD state_dict()s load into torch.optim.Adam / SGD built over the ORACLE modules' parameters (the
    reference's ``optim.load_state_dict(expe_state['optimizer']['state_dict'])``, src/main_baseline.py:874), the next
    step of both sides agrees, and a torch state dict loads back"""
    from bsed_amd.engine import FlatAdam, FlatSGD
    from bsed_amd.models import CRNN, Predictor
    kw = dict(co.CRNN_KWARGS); kw["dropout"] = 0.0
    ocrnn, opred = co.CRNN(**kw), co.Predictor(**co.PREDICTOR_KWARGS)
    seeded.load_seeded(ocrnn, 5); seeded.load_seeded(opred, 6)
    crnn, pred = CRNN(**kw), Predictor(**co.PREDICTOR_KWARGS)
    crnn.load_state_dict(ocrnn.state_dict()); pred.load_state_dict(opred.state_dict())
    oparams = list(ocrnn.parameters()) + list(opred.parameters())
    names = ["crnn." + n for n in crnn.reference_param_names()] + ["pred." + n for n in pred.reference_param_names()]
    onames = ["crnn." + n.replace("cnn.cnn.", "cnn.", 1) for n, _ in ocrnn.named_parameters()] + \
             ["pred." + n for n, _ in opred.named_parameters()]
    assert names == onames                                   # same index space as the reference optimizer
    g = torch.Generator().manual_seed(0)
    for make_mine, make_torch in ((lambda: FlatAdam([crnn, pred], lr=1e-3), lambda: torch.optim.Adam(oparams, lr=1e-3)),
                                  (lambda: FlatSGD([crnn, pred], lr=1e-2, momentum=0.9, weight_decay=1e-4),
                                   lambda: torch.optim.SGD(oparams, lr=1e-2, momentum=0.9, weight_decay=1e-4, nesterov=True))):
        mine, ref = make_mine(), make_torch()
        def set_grads():
            for (n, p) in zip(names, oparams):
                p.grad = torch.randn(p.shape, generator=g) * 1e-2
                mod, key = (crnn, n[5:]) if n.startswith("crnn.") else (pred, n[5:])
                mod.P(key).grad.copy_(p.grad.cuda())
        set_grads(); mine.step()
        ref.load_state_dict(mine.state_dict())               # torch accepts the flat optimizer's state
        for p, n in zip(oparams, names):                      # bring the oracle parameters to the stepped values
            mod, key = (crnn, n[5:]) if n.startswith("crnn.") else (pred, n[5:])
            p.data.copy_(mod.P(key).detach().cpu())
        set_grads(); mine.step(); ref.step()
        for p, n in zip(oparams, names):
            mod, key = (crnn, n[5:]) if n.startswith("crnn.") else (pred, n[5:])
            assert float((mod.P(key).detach().cpu() - p.data).abs().max()) <= 2e-6 * float(p.data.abs().max()) + 1e-9, n
        back = make_mine()
        back.load_state_dict(ref.state_dict())                # and the torch state loads into a fresh flat optimizer
        a, b = (mine.m, back.m) if hasattr(mine, "m") else (mine.buf, back.buf)
        for x, y in zip(a, b):
            assert float((x - y).abs().max()) <= 1e-6 * float(x.abs().max()) + 1e-12
