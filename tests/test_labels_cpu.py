"""Host-side label indexing of the package (bsed_amd.labels) against the vectors produced by the reference's
own ManyHotEncoder (tests/golden/labels_kat.json) -- bit-exact integer frames."""
import json
import os

import numpy as np

from bsed_amd.labels import BIRD_LIST, ManyHotEncoder, find_contiguous_regions


def test_encode_matches_reference(golden_dir):
    cases = json.load(open(os.path.join(golden_dir, "labels_kat.json")))
    enc = ManyHotEncoder(BIRD_LIST, n_frames=313)
    for c in cases:
        y = enc.encode_strong_df(c["events"])
        assert float(y.sum()) == c["sum"] and y.sum(0).tolist() == c["col_sums"]
        for lab, (first, last) in c["first_last"].items():
            col = y[:, BIRD_LIST.index(lab)].nonzero()[0]
            assert (int(col.min()), int(col.max()) + 1) == (first, last)
        assert enc.encode_weak([e[2] for e in c["events"]]).tolist() == c["weak"]
    assert (enc.frame(3.279), enc.frame(4.463)) == (102, 140)   # SURVEY.md 8 a5 known answers
    assert (enc.frame(6.550), enc.frame(8.213)) == (205, 257)


def test_encode_accepts_dataframe_and_roundtrips():
    import pandas as pd
    enc = ManyHotEncoder(BIRD_LIST, n_frames=313)
    df = pd.DataFrame({"onset": [1.0, 2.0], "offset": [3.0, 9.99], "event_label": ["EATO", "BAWW"]})
    y = enc.encode_strong_df(df)
    assert sorted(enc.decode_strong(y)) == sorted([["EATO", enc.frame(1.0), enc.frame(3.0)],
                                                   ["BAWW", enc.frame(2.0), enc.frame(9.99)]])
    assert enc.decode_weak(enc.encode_weak(["EATO,BAWW"])) == ["EATO", "BAWW"]
    assert (enc.encode_weak("empty") == -1).all()
    assert enc.encode_strong_df([]).sum() == 0           # empty annotation
    e2 = ManyHotEncoder.load_state_dict(enc.state_dict())
    assert e2.labels == BIRD_LIST and e2.n_frames == 313


def test_contiguous_regions_edges():
    assert find_contiguous_regions(np.zeros(5)).shape == (0, 2)
    assert find_contiguous_regions(np.ones(5)).tolist() == [[0, 5]]
    assert find_contiguous_regions(np.array([1, 0, 1, 1, 0, 1])).tolist() == [[0, 1], [2, 4], [5, 6]]
    # 22.05 kHz measurement config: frames follow the same float64 floor-division rule
    enc = ManyHotEncoder(BIRD_LIST, n_frames=216, sr=22050)
    assert enc.frame(9.99) == int(9.99 * 22050 // 255 // 4) == 215
