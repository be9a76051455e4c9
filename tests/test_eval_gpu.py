"""get_predictions (eval forward on the GPU + reference post-processing) against the oracle."""
import numpy as np
import pytest
import torch

from oracle import crnn_oracle as co
from oracle import labels_oracle as lo
from oracle import seeded

pytestmark = pytest.mark.gpu


def test_get_predictions_event_lists_and_feature_dump(tmp_path):
    from bsed_amd.evaluation import get_predictions
    from bsed_amd.labels import BIRD_LIST, ManyHotEncoder
    from bsed_amd.models import CRNN, Predictor
    seed, B, T = 31, 3, 256
    ocrnn, opred = co.CRNN(**co.CRNN_KWARGS), co.Predictor(**co.PREDICTOR_KWARGS)
    seeded.load_seeded(ocrnn, seed); seeded.load_seeded(opred, seed + 1)
    with torch.no_grad():
        opred.dense.bias += 1.0  # push some classes over the 0.5 threshold
    crnn, pred = CRNN(**co.CRNN_KWARGS), Predictor(**co.PREDICTOR_KWARGS)
    crnn.load_state_dict(ocrnn.state_dict()); pred.load_state_dict(opred.state_dict())
    x = seeded.db_like_input(seed + 2, B, T)
    # the reference's layout: <root>/wav/<name>.npy with <root>/annotation/<name>.txt next to it
    root = tmp_path / "d"
    (root / "wav").mkdir(parents=True); (root / "annotation").mkdir()
    truth = {0: [(1.0, 3.5, "EATO"), (4.0, 4.5, "AMCR")], 1: [], 2: [(0.0, 9.9, "BAWW")]}
    for i in range(B):
        with open(root / "annotation" / f"clip{i}.txt", "w") as f:
            f.write("onset\toffset\tevent_label\n")
            for on, off, lab in truth[i]:
                f.write(f"{on}\t{off}\t{lab}\n")
    loader = [(((torch.from_numpy(x), torch.from_numpy(x)), None), [str(root / "wav" / f"clip{i}.npy") for i in range(B)])]
    enc = ManyHotEncoder(BIRD_LIST, n_frames=T // 4)
    feat_dir = tmp_path / "feat"; feat_dir.mkdir()
    df, gt_df, dur_df = get_predictions(crnn, loader, enc.decode_strong, pooling_time_ratio=4, thresholds=[0.5],
                                        median_window=5, predictor=pred, saved_feature_dir=str(feat_dir),
                                        save_predictions=str(tmp_path / "pred.tsv"))
    # the reference's three return values (src/evaluation_measures.py:226-247,283)
    assert list(dur_df.columns) == ["filename", "duration"] and list(dur_df.filename) == ["clip0", "clip1", "clip2"]
    assert (dur_df.duration == 10).all()
    assert list(gt_df.columns) == ["onset", "offset", "event_label", "filename"]
    assert [(r.onset, r.offset, r.event_label, r.filename) for r in gt_df.itertuples()] == \
        [(1.0, 3.5, "EATO", "clip0"), (4.0, 4.5, "AMCR", "clip0"), (0.0, 9.9, "BAWW", "clip2")]
    ocrnn.eval(); opred.eval()
    with torch.no_grad():
        e, _ = ocrnn(torch.from_numpy(x))
        strong, _ = opred(e)
    ref = []
    for j in range(B):
        for lab, on, off in lo.post_process(strong[j].numpy(), median_window=5):
            ref.append((lab, round(on, 6), round(off, 6), f"clip{j}"))
    got = [(r.event_label, round(r.onset, 6), round(r.offset, 6), r.filename) for r in df.itertuples()]
    assert len(ref) > 0 and got == ref        # same rows in the reference's order: clip, class, time
    # a plain function as decoder takes the host path and must give the same list
    df2, _, _ = get_predictions(crnn, loader, lambda m: enc.decode_strong(m), pooling_time_ratio=4, thresholds=[0.5],
                                median_window=5, predictor=pred)
    assert [(r.event_label, r.onset, r.offset, r.filename) for r in df2.itertuples()] == \
        [(r.event_label, r.onset, r.offset, r.filename) for r in df.itertuples()]
    dumped = np.load(feat_dir / "0.npy")
    np.testing.assert_allclose(dumped, e.numpy(), atol=1e-4)
    assert (tmp_path / "pred.tsv").exists()
    assert crnn.training and pred.training  # restored


def test_decode_regions_gpu_matches_decode_strong_bit_exactly():
    """contiguous regions + seconds on the GPU against ManyHotEncoder.decode_strong / the reference's float64
    conversion, including columns that start on, end on, are empty, are full, and single-frame runs"""
    from bsed_amd.evaluation import decode_regions_gpu
    rng = np.random.default_rng(3)
    for B, T, C in ((3, 313, 20), (2, 216, 20), (5, 7, 3), (1, 1, 20)):
        m = (rng.random((B, T, C)) < 0.3).astype(np.float32)
        m[0, :, 0] = 1.0
        m[0, :, min(1, C - 1)] = 0.0 if C > 1 else 1.0
        if T > 2 and C > 2:
            m[-1, 0, 2] = 1.0; m[-1, 1, 2] = 0.0; m[-1, -1, 2] = 1.0
        scale = 4 / (32000 / 255)
        ev_clip, ev_class, ev_frames, ev_sec = decode_regions_gpu(torch.from_numpy(m).cuda(), scale, 10.0)
        want = [(b, c, int(on), int(off)) for b in range(B) for c in range(C)
                for on, off in lo.find_contiguous_regions(m[b, :, c])]
        got = [(int(b), int(c), int(f[0]), int(f[1])) for b, c, f in zip(ev_clip, ev_class, ev_frames)]
        assert got == want, (B, T, C)
        sec = np.clip(np.asarray([[w[2], w[3]] for w in want], dtype=np.float64).reshape(-1, 2) * scale, 0, 10.0)
        assert ev_sec.dtype == np.float64 and np.array_equal(ev_sec, sec)


def test_get_predictions_learned_post_unlabelled_clips_and_self_contained_model(tmp_path):
    """the reference's remaining call forms (src/evaluation_measures.py:163-199,226-247): class-wise median windows
    (learned_post, restated with scipy per class like the reference's loop), clips without an annotation file
    (unlabelled / pseudo-labelled sets), predictor=None with a model that returns (strong, weak) itself, empty decode"""
    import scipy.ndimage
    from bsed_amd.evaluation import (classwise_median_windows, decode_regions_gpu, get_predictions)
    from bsed_amd.labels import BIRD_LIST, ManyHotEncoder
    from bsed_amd.models import CRNN, Predictor
    seed, B, T = 33, 3, 256
    ocrnn, opred = co.CRNN(**co.CRNN_KWARGS), co.Predictor(**co.PREDICTOR_KWARGS)
    seeded.load_seeded(ocrnn, seed); seeded.load_seeded(opred, seed + 1)
    with torch.no_grad():
        opred.dense.bias += 1.0
    crnn, pred = CRNN(**co.CRNN_KWARGS), Predictor(**co.PREDICTOR_KWARGS)
    crnn.load_state_dict(ocrnn.state_dict()); pred.load_state_dict(opred.state_dict())
    x = seeded.db_like_input(seed + 2, B, T)
    root = tmp_path / "d"
    (root / "wav").mkdir(parents=True); (root / "annotation").mkdir()
    with open(root / "annotation" / "clip1.txt", "w") as f:          # only ONE of the three clips is labelled
        f.write("onset\toffset\tevent_label\n2.0\t3.0\tEATO\n")
    loader = [(((torch.from_numpy(x), torch.from_numpy(x)), None), [str(root / "wav" / f"clip{i}.npy") for i in range(B)])]
    enc = ManyHotEncoder(BIRD_LIST, n_frames=T // 4)
    windows = classwise_median_windows(32000, 255, 4)
    assert windows == [14, 14, 14, 14, 14, 84, 84, 84, 14, 84]       # cfg.median_window of the reference at 32 kHz / 255 / 4
    df, gt_df, dur_df = get_predictions(crnn, loader, enc.decode_strong, pooling_time_ratio=4, thresholds=[0.5],
                                        predictor=pred, learned_post=True)
    assert list(gt_df.filename) == ["clip1"] and len(dur_df) == 3
    with pytest.raises(FileNotFoundError):
        get_predictions(crnn, loader, enc.decode_strong, pooling_time_ratio=4, predictor=pred, require_annotations=True)
    ocrnn.eval(); opred.eval()
    with torch.no_grad():
        strong, _ = opred(ocrnn(torch.from_numpy(x))[0])
    ref = []
    scale = 4 / (32000 / 255)
    for j in range(B):
        binar = (strong[j].numpy() > 0.5).astype(np.float64)
        cols = [scipy.ndimage.median_filter(binar[:, k:k + 1], (windows[k], 1)) for k in range(len(windows))]
        m = np.hstack(cols)                                         # 10 columns: the classes beyond the list drop out
        for k in range(m.shape[1]):
            for on, off in lo.find_contiguous_regions(m[:, k]):
                ref.append((BIRD_LIST[k], round(float(np.clip(on * scale, 0, 10)), 6),
                            round(float(np.clip(off * scale, 0, 10)), 6), f"clip{j}"))
    got = [(r.event_label, round(r.onset, 6), round(r.offset, 6), r.filename) for r in df.itertuples()]
    assert len(ref) > 0 and got == ref

    class Whole(torch.nn.Module):                                     # a model that carries its own head (reference :180-181)
        def forward(self, inp, inference=False):
            return pred(crnn(inp)[0], inference=inference)
    whole = Whole()
    df_a, gt_a, _ = get_predictions(whole, loader, enc.decode_strong, pooling_time_ratio=4, median_window=5, fpn=True)
    df_b, _, _ = get_predictions(crnn, loader, enc.decode_strong, pooling_time_ratio=4, median_window=5, fpn=True,
                                 predictor=pred)
    assert len(df_b) > 0 and df_a.equals(df_b)
    with pytest.raises(NotImplementedError):
        get_predictions(whole, loader, enc.decode_strong)
    # no clip with an annotation at all -> groundtruth_df None; an empty batch decodes to an empty list
    (root / "annotation" / "clip1.txt").unlink()
    assert get_predictions(crnn, loader, enc.decode_strong, pooling_time_ratio=4, predictor=pred)[1] is None
    out = decode_regions_gpu(torch.zeros((0, 64, 20), device="cuda"), scale, 10.0)
    assert all(len(a) == 0 for a in out)
