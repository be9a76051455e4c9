"""On-disk formats and loaders of the reference, feeding the GPU path unchanged.

  wav/<name>.npy         float32 (T,128) LINEAR mel amplitude   (written by ena_data_preprocess / syn_preprocess,
  annotation/<name>.txt  TSV onset / offset / event_label        reference src/data/preprocess.py:204-229)
  FeatureDataset         <- ENA_Dataset / SYN_Dataset, reference src/data/dataload.py:17-82,127-196
                            (item = ((features, target), path); the per-item CPU transforms of the reference --
                             noise, amplitude_to_db, pad/trunc -- are NOT applied here)
  GpuCollate             batches items and runs the reference's transform chain (get_transforms,
                            src/data/Transforms.py:304-322) for the WHOLE batch on the GPU (csrc/mel.hip kernels):
                            returns ((x, x_noisy), target), paths like the reference's DataLoader batches.
  write_features         preprocess() + the reference's file layout (np.save + to_csv(sep="\\t"))
"""
import glob
import os

import numpy as np
import torch


def read_annotation(path):
    """rows of (onset, offset, event_label) from the reference's TSV (header: onset offset event_label ...)"""
    import pandas as pd
    df = pd.read_csv(path, sep="\t")
    return df


class FeatureDataset(torch.utils.data.Dataset):
    def __init__(self, preprocess_dir, encod_func, transform=None):
        self.preprocess_dir = preprocess_dir
        self.annotation_dir = os.path.join(preprocess_dir, "annotation")
        self.feature_dir = os.path.join(preprocess_dir, "wav")
        self.feature_file_list = sorted(glob.glob(os.path.join(self.feature_dir, "*.npy")))
        self.encod_func, self.transform = encod_func, transform

    def __len__(self):
        return len(self.feature_file_list)

    def __getitem__(self, index):
        path = self.feature_file_list[index]
        features = np.load(path)
        name = os.path.splitext(os.path.basename(path))[0]
        df = read_annotation(os.path.join(self.annotation_dir, name + ".txt"))
        target = self.encod_func(df)
        sample = (features, target)
        if self.transform is not None:
            sample = self.transform(sample)
        return sample, path


class GpuCollate:
    """collate_fn: list of ((linear_mel (T,128), target), path) -> (((x, x_noisy), target), paths) with
    x, x_noisy = (B,1,max_frames,128) dB-mel on the GPU (clean / SNR-noise view), target float32 on the GPU."""

    def __init__(self, frontend, max_frames=None, noisy=True, seed=0):
        self.fe, self.max_frames, self.noisy, self.seed = frontend, max_frames, noisy, seed
        self.calls = 0

    def __call__(self, items):
        feats = [np.asarray(s[0], dtype=np.float32) for s, _ in items]
        paths = [p for _, p in items]
        T = max(f.shape[0] for f in feats)
        max_frames = self.fe.cfg.max_frames if self.max_frames is None else self.max_frames
        B, M = len(feats), feats[0].shape[1]
        out_c, out_n = [], []
        # clips of equal length share one launch; ragged clips (shorter final segments) go one by one so that each
        # keeps its own per-clip top_db clamp and noise statistics, exactly like the per-item reference transform
        groups = {}
        for i, f in enumerate(feats):
            groups.setdefault(f.shape[0], []).append(i)
        clean = torch.empty((B, 1, max_frames, M), device="cuda", dtype=torch.float32)
        noisy = torch.empty_like(clean) if self.noisy else None
        for Tlen, idx in groups.items():
            mel = torch.from_numpy(np.stack([feats[i] for i in idx])).cuda()
            cmax, sumsq = self.fe.stats(mel)
            c = self.fe.to_db(mel, cmax, max_frames)
            clean[idx] = c
            if self.noisy:
                nz, nmax = self.fe.add_noise(mel, sumsq, seed=self.seed + self.calls)
                noisy[idx] = self.fe.to_db(nz, nmax, max_frames)
        self.calls += 1
        target = torch.from_numpy(np.stack([np.asarray(s[1], dtype=np.float32) for s, _ in items])).cuda()
        return ((clean, noisy if self.noisy else clean), target), paths


def write_features(out_dir, name, audio, events, frontend=None):
    """<out_dir>/wav/<name>.npy + <out_dir>/annotation/<name>.txt in the reference's layout.
    events: iterable of (onset_s, offset_s, label)."""
    import pandas as pd
    from .features import preprocess
    os.makedirs(os.path.join(out_dir, "wav"), exist_ok=True)
    os.makedirs(os.path.join(out_dir, "annotation"), exist_ok=True)
    mel = preprocess(audio, cfg=frontend.cfg if frontend is not None else None)
    np.save(os.path.join(out_dir, "wav", name + ".npy"), mel)
    df = pd.DataFrame([{"onset": a, "offset": b, "event_label": c} for a, b, c in events],
                      columns=["onset", "offset", "event_label"])
    df.to_csv(os.path.join(out_dir, "annotation", name + ".txt"), sep="\t", index=False)
    return mel
