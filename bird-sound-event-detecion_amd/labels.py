"""Frame/time label indexing -- host-side integer logic, bit-exact with the reference.

  ManyHotEncoder   <- reference src/utilities/ManyHotEncoder.py:7-173 (same constructor, method names and
                      state_dict; the module-level ``cfg`` constants become constructor arguments)
The index math ``int(t * sr // hop // pooling)`` is float64 floor division and is reproduced as such
(reference :121-122); ``decode_strong`` restates dcase_util's ``find_contiguous_regions``.
"""
import numpy as np

BIRD_LIST = ["EATO", "WOTH", "BCCH", "BTNW", "TUTI", "NOCA", "REVI", "AMCR", "BLJA", "OVEN",
             "COYE", "BGGN", "SCTA", "AMRE", "KEWA", "BHCO", "BHVI", "HETH", "RBWO", "BAWW"]


def find_contiguous_regions(activity_array):
    a = np.asarray(activity_array).astype(bool)
    change = np.logical_xor(a[1:], a[:-1]).nonzero()[0] + 1
    if a.size and a[0]:
        change = np.r_[0, change]
    if a.size and a[-1]:
        change = np.r_[change, a.size]
    return change.reshape((-1, 2))


class ManyHotEncoder:
    def __init__(self, labels, n_frames=None, sr=32000, hop_size=255, pooling_time_ratio=4):
        if isinstance(labels, np.ndarray):
            labels = labels.tolist()
        self.labels = list(labels)
        self.n_frames = n_frames
        self.sr, self.hop_size, self.pooling_time_ratio = sr, hop_size, pooling_time_ratio

    def frame(self, seconds):
        return int(float(seconds) * self.sr // self.hop_size // self.pooling_time_ratio)

    def encode_weak(self, labels):
        if isinstance(labels, str):
            if labels == "empty":
                return np.zeros(len(self.labels)) - 1
            labels = [labels]
        if hasattr(labels, "columns"):  # pandas DataFrame
            labels = [] if labels.empty else (labels["event_label"] if "event_label" in labels.columns else labels)
        y = np.zeros(len(self.labels))
        for label in labels:
            for event in str(label).split(","):
                if event == "nan":
                    continue
                y[self.labels.index(event)] = 1
        return y

    def encode_strong_df(self, label_df):
        """rows of (onset [s], offset [s], event_label): DataFrame, or iterable of such triples / dicts"""
        y = np.zeros((self.n_frames, len(self.labels)))
        if hasattr(label_df, "iterrows"):
            rows = ((r["onset"], r["offset"], r["event_label"]) for _, r in label_df.iterrows())
        else:
            rows = ((r["onset"], r["offset"], r["event_label"]) if isinstance(r, dict) else r for r in label_df)
        for onset, offset, label in rows:
            i = self.labels.index(label)
            y[self.frame(onset):self.frame(offset), i] = 1
        return y

    def decode_weak(self, labels):
        return [self.labels[i] for i, v in enumerate(labels) if v == 1]

    def decode_strong(self, labels):
        result = []
        for i, col in enumerate(np.asarray(labels).T):
            for on, off in find_contiguous_regions(col):
                result.append([self.labels[i], int(on), int(off)])
        return result

    def state_dict(self):
        return {"labels": self.labels, "n_frames": self.n_frames}

    @classmethod
    def load_state_dict(cls, state_dict, **kw):
        return cls(state_dict["labels"], state_dict["n_frames"], **kw)
