"""On-disk formats and loaders of the reference, feeding the GPU path unchanged.

  wav/<name>.npy         float32 (T,128) LINEAR mel amplitude   (written by ena_data_preprocess / syn_preprocess,
  annotation/<name>.txt  TSV onset / offset / event_label        reference src/data/preprocess.py:204-229)
  FeatureDataset         <- ENA_Dataset / SYN_Dataset, reference src/data/dataload.py:17-82,127-196
                            (item = ((features, target), path); the per-item CPU transforms of the reference --
                             noise, amplitude_to_db, pad/trunc -- are NOT applied here)
  PseudoWeakDataset      <- ENA_Dataset_unlabeled, reference src/data/dataload.py:84-126: unlabeled in-domain clips whose
                            weak targets come from ONE pseudo-label TSV (``filename<TAB>event_labels``, the shipped
                            src/unlabel_in_domain_pseudo_weak*.tsv) instead of per-clip annotation files
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


class PseudoWeakDataset(torch.utils.data.Dataset):
    """Unlabeled in-domain clips with pseudo weak labels (reference ``ENA_Dataset_unlabeled``, dataload.py:84-126).

    item = ((features, target), path) with ``target = encod_func(event_labels of the rows whose filename == path)`` --
    a pandas Series of comma-separated label strings, which is what ``ManyHotEncoder.encode_weak`` takes; a clip without
    a row gets the all-zero vector, as in the reference.  The reference hard-codes the TSV's location and re-reads it for
    every item; here it is an argument and read once.  ``match``: "path" compares the full feature path with the TSV's
    ``filename`` column exactly like the reference; "basename" compares file names only (for a data set that was moved
    after the TSV was written -- the shipped files carry /home/fumchin/... paths)."""

    def __init__(self, preprocess_dir, encod_func, transform=None, pseudo_label_tsv=None, match="path", compute_log=False):
        import pandas as pd
        if pseudo_label_tsv is None:
            raise ValueError("PseudoWeakDataset needs pseudo_label_tsv (the reference hard-codes "
                             "src/unlabel_in_domain_pseudo_weak_resNet.tsv)")
        if match not in ("path", "basename"):
            raise ValueError("match must be 'path' or 'basename'")
        self.preprocess_dir = preprocess_dir
        self.annotation_dir = pseudo_label_tsv              # (the reference keeps the TSV path under this name)
        self.feature_dir = os.path.join(preprocess_dir, "wav")
        self.feature_file_list = sorted(glob.glob(os.path.join(self.feature_dir, "*.npy")))
        self.encod_func, self.transform, self.match = encod_func, transform, match
        df = pd.read_csv(pseudo_label_tsv, sep="\t")
        if not {"filename", "event_labels"}.issubset(df.columns):
            raise ValueError(f"{pseudo_label_tsv}: expected the columns filename and event_labels, got {list(df.columns)}")
        self._key = df["filename"] if match == "path" else df["filename"].map(os.path.basename)
        self._df = df

    def __len__(self):
        return len(self.feature_file_list)

    def labels_of(self, path):
        key = path if self.match == "path" else os.path.basename(path)
        return self._df[self._key == key]["event_labels"]

    def __getitem__(self, index):
        path = self.feature_file_list[index]
        features = np.load(path)
        target = self.encod_func(self.labels_of(path))
        sample = (features, target)
        if self.transform is not None:
            sample = self.transform(sample)
        return sample, path


# the reference's class names (src/data/dataload.py), for import-swap drivers
ENA_Dataset = FeatureDataset
SYN_Dataset = FeatureDataset
ENA_Dataset_unlabeled = PseudoWeakDataset


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
