"""Frame/time indexing oracle.  TEST INFRASTRUCTURE ONLY -- never on the product path.

  * encode_strong / encode_weak  <- /root/reference/src/utilities/ManyHotEncoder.py:27-54,
                                    56-130 (index math :121-122, float64 floor-div)
  * find_contiguous_regions      <- dcase_util DecisionEncoder.find_contiguous_regions
                                    (third-party, unpinned; published algorithm: xor of the
                                    shifted boolean vector -> change indices -> pairs)
  * decode_strong                <- ManyHotEncoder.py:148-164
  * post_process                 <- src/evaluation_measures.py:184-215 (threshold 0.5, median
                                    filter (w,1), decode, frames -> seconds, clip [0, 10])

Pinned by the known-answer rows of the reference's dataset/SYN/generated/output.tsv
(tests/golden/labels_kat.json).
"""
import numpy as np
import scipy.ndimage

BIRD_LIST = ["EATO", "WOTH", "BCCH", "BTNW", "TUTI", "NOCA", "REVI", "AMCR", "BLJA", "OVEN",
             "COYE", "BGGN", "SCTA", "AMRE", "KEWA", "BHCO", "BHVI", "HETH", "RBWO", "BAWW"]


def frame_index(t_seconds, sr=32000, hop=255, pooling=4):
    """int(t * sr // hop // pooling) with python-float (float64) floor division."""
    return int(float(t_seconds) * sr // hop // pooling)


def encode_strong(events, n_frames, labels=BIRD_LIST, sr=32000, hop=255, pooling=4):
    """events: iterable of (onset_s, offset_s, label) -> float64 (n_frames, n_class)."""
    y = np.zeros((n_frames, len(labels)))
    for on, off, lab in events:
        i = labels.index(lab) if isinstance(lab, str) else int(lab)
        y[frame_index(on, sr, hop, pooling):frame_index(off, sr, hop, pooling), i] = 1
    return y


def encode_weak(event_labels, labels=BIRD_LIST):
    y = np.zeros(len(labels))
    for lab in event_labels:
        for ev in lab.split(","):
            y[labels.index(ev)] = 1
    return y


def find_contiguous_regions(activity):
    a = np.asarray(activity).astype(bool)
    change = np.logical_xor(a[1:], a[:-1]).nonzero()[0] + 1
    if a[0]:
        change = np.r_[0, change]
    if a[-1]:
        change = np.r_[change, a.size]
    return change.reshape((-1, 2))


def decode_strong(y, labels=BIRD_LIST):
    out = []
    for i, col in enumerate(np.asarray(y).T):
        for on, off in find_contiguous_regions(col):
            out.append([labels[i], int(on), int(off)])
    return out


def post_process(strong, threshold=0.5, median_window=14, pooling=4, sr=32000, hop=255,
                 max_len_seconds=10.0, labels=BIRD_LIST):
    """(T', n_class) probabilities -> [[label, onset_s, offset_s], ...]."""
    binar = (np.asarray(strong) > threshold).astype(np.float64)
    binar = scipy.ndimage.median_filter(binar, (median_window, 1))
    ev = decode_strong(binar, labels)
    scale = pooling / (sr / hop)
    return [[l, float(np.clip(on * scale, 0, max_len_seconds)),
             float(np.clip(off * scale, 0, max_len_seconds))] for l, on, off in ev]
