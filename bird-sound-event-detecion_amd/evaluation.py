"""Inference / event decoding -- host-side mirror of the reference's ``get_predictions``.

  get_predictions   <- reference src/evaluation_measures.py:123-283: eval-mode forward (HIP), threshold,
                       (median_window, 1) median filter, ``decoder`` (ManyHotEncoder.decode_strong), frames ->
                       seconds with ``pooling_time_ratio / (sr / hop)``, clip to [0, max_len_seconds];
                       optional embedding dump ``<saved_feature_dir>/<i>.npy`` (what save_features.py is for).
Metric values (sed_eval / psds_eval) stay external: only the event lists are produced here.
The post-processing runs on the host exactly as in the reference (pandas / scipy); moving it to the GPU is
SURVEY.md section 8(f) rank 3.
"""
import os

import numpy as np
import scipy.ndimage
import torch


def post_process(pred_strong, decoder, threshold=0.5, median_window=1, pooling_time_ratio=1, sr=32000,
                 hop_size=255, max_len_seconds=10.0):
    """(T', C) probabilities -> list of [event_label, onset_s, offset_s]"""
    binar = (np.asarray(pred_strong) > threshold).astype(np.float64)
    binar = scipy.ndimage.median_filter(binar, (median_window, 1))
    scale = pooling_time_ratio / (sr / hop_size)
    return [[lab, float(np.clip(on * scale, 0, max_len_seconds)), float(np.clip(off * scale, 0, max_len_seconds))]
            for lab, on, off in decoder(binar)]


def binarize_median_gpu(pred_strong, threshold=0.5, median_window=1):
    """(B,T',C) GPU probabilities -> (B,T',C) 0/1 mask: threshold + scipy-compatible median filter, one HIP kernel
    for the whole batch instead of a per-clip scipy call"""
    import ctypes
    from . import _lib as L
    x = pred_strong.contiguous()
    B, T, C = x.shape
    out = torch.empty_like(x)
    L.call("bsed_binarize_median", L.ptr(x), L.ptr(out), L.c_int(B), L.c_int(T), L.c_int(C), ctypes.c_float(threshold),
           L.c_int(median_window), L.stream())
    return out


def get_predictions(model, dataloader, decoder, pooling_time_ratio=1, thresholds=(0.5,), median_window=1,
                    save_predictions=None, del_model=False, learned_post=False, predictor=None, fpn=False,
                    saved_feature_dir=None, sr=32000, hop_size=255, max_len_seconds=10.0):
    """Same call signature as the reference.  ``dataloader`` yields
    ``(((input, ema_input), target), paths)`` batches; returns a DataFrame (or list per threshold) with
    columns event_label / onset / offset / filename (seconds).  Ground-truth and duration frames are built
    by the caller's own annotation reader (the reference reads ``annotation/<name>.txt`` next to the features)."""
    import pandas as pd
    if predictor is None:
        raise NotImplementedError("bsed_amd.get_predictions needs the CRNN + Predictor pair (predictor=...)")
    if learned_post:
        raise NotImplementedError("learned_post (class-wise median windows) is not on the hot path")
    was_training = (model.training, predictor.training)
    model.eval(); predictor.eval()
    rows = {t: [] for t in thresholds}
    for i, (((input_data, _ema), _target), paths) in enumerate(dataloader):
        names = [os.path.splitext(os.path.basename(p))[0] for p in paths]
        with torch.no_grad():
            x = torch.as_tensor(input_data).float().cuda()
            encoded_x, feature_out = model(x)
            pred_strong, _ = predictor(encoded_x, inference=fpn)
        if saved_feature_dir is not None:
            np.save(os.path.join(saved_feature_dir, f"{i}"), feature_out.cpu().numpy())
        scale = pooling_time_ratio / (sr / hop_size)
        for t in thresholds:
            # threshold + median filter for the whole batch on the GPU; only the 0/1 masks travel to the host
            masks = binarize_median_gpu(pred_strong, t, median_window).cpu().numpy()
            for j, m in enumerate(masks):
                for lab, on, off in decoder(m):
                    rows[t].append({"event_label": lab, "onset": float(np.clip(on * scale, 0, max_len_seconds)),
                                    "offset": float(np.clip(off * scale, 0, max_len_seconds)), "filename": names[j]})
    model.train(was_training[0]); predictor.train(was_training[1])
    dfs = [pd.DataFrame(rows[t], columns=["event_label", "onset", "offset", "filename"]) for t in thresholds]
    if save_predictions is not None:
        outs = [save_predictions] if isinstance(save_predictions, str) and len(dfs) == 1 else save_predictions
        if isinstance(outs, str):
            base, ext = os.path.splitext(outs)
            outs = [os.path.join(base, f"{t:.3f}{ext}") for t in thresholds]
        for df, path in zip(dfs, outs):
            if os.path.dirname(path):
                os.makedirs(os.path.dirname(path), exist_ok=True)
            df.to_csv(path, index=False, sep="\t", float_format="%.3f")
    return dfs[0] if len(dfs) == 1 else dfs
