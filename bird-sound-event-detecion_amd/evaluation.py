"""Inference / event decoding -- host-side mirror of the reference's ``get_predictions``.

  get_predictions   <- reference src/evaluation_measures.py:123-283: eval-mode forward, threshold, (median_window, 1)
                       median filter, ``decoder`` (ManyHotEncoder.decode_strong: contiguous regions), frames -> seconds
                       with ``pooling_time_ratio / (sr / hop)`` clipped to [0, max_len_seconds]; ground-truth frame
                       from ``annotation/<name>.txt`` next to the features, duration frame (10 s per clip); optional
                       embedding dump ``<saved_feature_dir>/<i>.npy`` (what save_features.py is for).
                       Returns ``(predictions, groundtruth_df, duration_df)`` like the reference (:283).
Everything per frame runs on the GPU for the whole batch: forward (HIP), threshold + median filter
(``bsed_binarize_median``), contiguous-region decode and the seconds conversion (``bsed_decode_count`` /
``bsed_decode_write``).  Only the event list (a few rows per clip) travels to the host, where the DataFrames are
assembled without a per-clip Python loop.  Metric values (sed_eval / psds_eval) stay external.
"""
import ctypes
import os

import numpy as np
import torch

from . import _lib as L


def post_process(pred_strong, decoder, threshold=0.5, median_window=1, pooling_time_ratio=1, sr=32000,
                 hop_size=255, max_len_seconds=10.0):
    """Host restatement for ONE clip ((T', C) probabilities -> list of [event_label, onset_s, offset_s]); kept for
    callers that hand in a custom ``decoder`` function (the GPU decode needs the label list of a ManyHotEncoder)."""
    import scipy.ndimage
    binar = (np.asarray(pred_strong) > threshold).astype(np.float64)
    binar = scipy.ndimage.median_filter(binar, (median_window, 1))
    scale = pooling_time_ratio / (sr / hop_size)
    return [[lab, float(np.clip(on * scale, 0, max_len_seconds)), float(np.clip(off * scale, 0, max_len_seconds))]
            for lab, on, off in decoder(binar)]


def binarize_median_gpu(pred_strong, threshold=0.5, median_window=1):
    """(B,T',C) GPU probabilities -> (B,T',C) 0/1 mask: threshold + scipy-compatible median filter, one HIP kernel
    for the whole batch instead of a per-clip scipy call"""
    x = pred_strong.contiguous()
    B, T, C = x.shape
    out = torch.empty_like(x)
    L.call("bsed_binarize_median", L.ptr(x), L.ptr(out), L.c_int(B), L.c_int(T), L.c_int(C), ctypes.c_float(threshold),
           L.c_int(median_window), L.stream())
    return out


def decode_regions_gpu(mask, scale, max_len_seconds):
    """(B,T',C) 0/1 GPU mask -> (clip (E,), class (E,), frames (E,2), seconds (E,2)) numpy arrays, ordered by clip,
    class, time: ManyHotEncoder.decode_strong for every clip of the batch plus the frames -> seconds conversion, in
    two HIP launches (count, write at the exclusive prefix of the counts)."""
    mask = mask.contiguous()
    B, T, C = mask.shape
    if B * C == 0 or T == 0:                            # an empty batch decodes to an empty event list
        return (np.zeros(0, np.int32), np.zeros(0, np.int32), np.zeros((0, 2), np.int32), np.zeros((0, 2), np.float64))
    counts = torch.empty(B * C, device=mask.device, dtype=torch.int32)
    L.call("bsed_decode_count", L.ptr(mask), L.c_int(B), L.c_int(T), L.c_int(C), L.ptr(counts, torch.int32), L.stream())
    csum = torch.cumsum(counts, 0, dtype=torch.int32)
    offsets = (csum - counts).contiguous()
    E = int(csum[-1])                                   # the one host sync of the decode: the list length
    ev_clip = torch.empty(max(E, 1), device=mask.device, dtype=torch.int32)
    ev_class = torch.empty(max(E, 1), device=mask.device, dtype=torch.int32)
    ev_frames = torch.empty((max(E, 1), 2), device=mask.device, dtype=torch.int32)
    ev_seconds = torch.empty((max(E, 1), 2), device=mask.device, dtype=torch.float64)
    if E:
        L.call("bsed_decode_write", L.ptr(mask), L.ptr(offsets, torch.int32), L.c_int(B), L.c_int(T), L.c_int(C),
               ctypes.c_double(scale), ctypes.c_double(max_len_seconds), L.ptr(ev_clip, torch.int32),
               L.ptr(ev_class, torch.int32), L.ptr(ev_frames, torch.int32), L.ptr(ev_seconds, torch.float64), L.stream())
    return (ev_clip[:E].cpu().numpy(), ev_class[:E].cpu().numpy(), ev_frames[:E].cpu().numpy(),
            ev_seconds[:E].cpu().numpy())


def _decoder_labels(decoder):
    """label list of the ManyHotEncoder whose bound ``decode_strong`` was passed as ``decoder`` (the reference's call
    sites pass ``many_hot_encoder.decode_strong``, src/main_baseline.py:1010-1032), or None for any other callable"""
    owner = getattr(decoder, "__self__", None)
    if owner is not None and getattr(decoder, "__name__", "") == "decode_strong" and hasattr(owner, "labels"):
        return list(owner.labels)
    return None


# reference src/data/config.py:62-63: cfg.median_window = [max(int(s * out_nb_frames_1s), 1) for s in median_window_s_classwise]
MEDIAN_WINDOW_S_CLASSWISE = [0.45, 0.45, 0.45, 0.45, 0.45, 2.7, 2.7, 2.7, 0.45, 2.7]


def classwise_median_windows(sr=32000, hop_size=255, pooling_time_ratio=4, seconds=MEDIAN_WINDOW_S_CLASSWISE):
    out_nb_frames_1s = sr / hop_size / pooling_time_ratio
    return [max(int(s * out_nb_frames_1s), 1) for s in seconds]


def binarize_median_classwise_gpu(pred_strong, threshold, windows):
    """``learned_post`` of the reference (src/evaluation_measures.py:192-197): class k gets its own median window
    ``windows[k]``; like the reference's ``np.hstack`` over ``range(len(cfg.median_window))``, classes beyond the list
    are dropped (no events).  One HIP launch per DISTINCT window, columns merged on the GPU."""
    B, T, C = pred_strong.shape
    out = torch.zeros_like(pred_strong)
    for w in sorted(set(windows[:C])):
        cols = torch.tensor([k for k, wk in enumerate(windows[:C]) if wk == w], device=pred_strong.device)
        out[:, :, cols] = binarize_median_gpu(pred_strong, threshold, w)[:, :, cols]
    return out


def get_predictions(model, dataloader, decoder, pooling_time_ratio=1, thresholds=(0.5,), median_window=1,
                    save_predictions=None, del_model=False, learned_post=False, predictor=None, fpn=False,
                    saved_feature_dir=None, sr=32000, hop_size=255, max_len_seconds=10.0, classwise_median_window=None,
                    require_annotations=False):
    """Same call signature and return value as the reference: ``(predictions, groundtruth_df, duration_df)``.
    ``dataloader`` yields ``(((input, ema_input), target), paths)`` batches.  predictions: one DataFrame (or a list,
    one per threshold) with columns event_label / onset / offset / filename (seconds); groundtruth_df: the
    ``annotation/<name>.txt`` files of the clips concatenated with a ``filename`` column; duration_df: filename /
    duration (10, as the reference hard-codes it).
    learned_post: class-wise median windows (``classwise_median_window``, default the reference's ``cfg.median_window``
    list for this sr / hop / pooling).  predictor=None: ``model`` returns ``(strong, weak)`` itself and is called as
    ``model(x, inference=True)`` when ``fpn`` (reference :180-181; its ``seg_index`` form belongs to a model class that is
    not on the path).  Clips without an ``annotation/<name>.txt`` (unlabelled / pseudo-labelled sets) are left out of
    groundtruth_df -- None when no clip has one -- unless ``require_annotations`` (the reference's behaviour: it raises)."""
    import pandas as pd
    if predictor is None and not fpn:
        raise NotImplementedError("get_predictions(predictor=None, fpn=False) is the reference's seg_index call of a model "
                                  "class outside the hot path; pass predictor=... or a self-contained model with fpn=True")
    if learned_post and classwise_median_window is None:
        classwise_median_window = classwise_median_windows(sr, hop_size, pooling_time_ratio)
    was_training = (model.training, predictor.training if predictor is not None else False)
    model.eval()
    if predictor is not None:
        predictor.eval()
    labels = _decoder_labels(decoder)
    scale = pooling_time_ratio / (sr / hop_size)
    frames = {t: [] for t in thresholds}
    filename_list, annotation_folder_list = [], []
    for i, (((input_data, _ema), _target), paths) in enumerate(dataloader):
        names = [os.path.splitext(os.path.basename(p))[0] for p in paths]
        folders = [os.path.join(os.path.dirname(os.path.dirname(p)), "annotation") for p in paths]
        with torch.no_grad():
            x = torch.as_tensor(input_data).float().cuda()
            if predictor is not None:
                encoded_x, feature_out = model(x)
                pred_strong, _ = predictor(encoded_x, inference=fpn)
            else:
                pred_strong, feature_out = model(x, inference=True)[0], None
        if saved_feature_dir is not None and feature_out is not None:
            np.save(os.path.join(saved_feature_dir, f"{i}"), feature_out.cpu().numpy())
        for t in thresholds:
            mask = (binarize_median_classwise_gpu(pred_strong, t, list(classwise_median_window)) if learned_post
                    else binarize_median_gpu(pred_strong, t, median_window))
            if labels is not None:
                ev_clip, ev_class, _, ev_sec = decode_regions_gpu(mask, scale, max_len_seconds)
                frames[t].append(pd.DataFrame({"event_label": np.asarray(labels, dtype=object)[ev_class],
                                               "onset": ev_sec[:, 0], "offset": ev_sec[:, 1],
                                               "filename": np.asarray(names, dtype=object)[ev_clip]}))
            else:
                # a caller-supplied decoder function can only run on the host, clip by clip
                rows = []
                for j, m in enumerate(mask.cpu().numpy()):
                    for lab, on, off in decoder(m):
                        rows.append({"event_label": lab, "onset": float(np.clip(on * scale, 0, max_len_seconds)),
                                     "offset": float(np.clip(off * scale, 0, max_len_seconds)), "filename": names[j]})
                frames[t].append(pd.DataFrame(rows, columns=["event_label", "onset", "offset", "filename"]))
        filename_list += names
        annotation_folder_list += folders
    model.train(was_training[0])
    if predictor is not None:
        predictor.train(was_training[1])
    cols = ["event_label", "onset", "offset", "filename"]
    dfs = [pd.concat(frames[t], ignore_index=True)[cols] if frames[t] else pd.DataFrame(columns=cols)
           for t in thresholds]

    # ground-truth and duration frames (reference :226-247): first occurrence of every file name, its annotation file
    # next to the features, duration 10
    seen = {}
    for name, folder in zip(filename_list, annotation_folder_list):
        seen.setdefault(name, folder)
    duration_df = pd.DataFrame(list(seen.keys()), columns=["filename"])
    duration_df["duration"] = 10
    groundtruth_df = None
    gts, n_found = [], 0
    for name, folder in seen.items():
        path = os.path.join(folder, name + ".txt")
        if not os.path.exists(path):
            if require_annotations:
                raise FileNotFoundError(f"get_predictions: annotation file {path} is missing (the reference reads "
                                        "annotation/<name>.txt next to wav/<name>.npy)")
            continue                                    # unlabelled clip: predictions only
        n_found += 1
        df = pd.read_csv(path, sep="\t")
        df["filename"] = name
        if len(df):
            gts.append(df)
    if gts:
        groundtruth_df = pd.concat(gts, ignore_index=True)
    elif n_found:
        groundtruth_df = pd.DataFrame(columns=["onset", "offset", "event_label", "filename"])

    if save_predictions is not None:
        if isinstance(save_predictions, str):
            if len(thresholds) == 1:
                outs = [save_predictions]
            else:
                base, ext = os.path.splitext(save_predictions)
                outs = [os.path.join(base, f"{t:.3f}{ext}") for t in thresholds]
        else:
            assert len(save_predictions) == len(thresholds), \
                f"There should be a prediction file per threshold: {len(save_predictions)} vs {len(thresholds)}"
            outs = list(save_predictions)
        for df, path in zip(dfs, outs):
            if os.path.dirname(path):
                os.makedirs(os.path.dirname(path), exist_ok=True)
            df.to_csv(path, index=False, sep="\t", float_format="%.3f")
    predictions = dfs[0] if len(dfs) == 1 else dfs
    return predictions, groundtruth_df, duration_df
