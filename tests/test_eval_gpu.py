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
    loader = [(((torch.from_numpy(x), torch.from_numpy(x)), None), [f"/d/wav/clip{i}.npy" for i in range(B)])]
    enc = ManyHotEncoder(BIRD_LIST, n_frames=T // 4)
    feat_dir = tmp_path / "feat"; feat_dir.mkdir()
    df = get_predictions(crnn, loader, enc.decode_strong, pooling_time_ratio=4, thresholds=[0.5], median_window=5,
                         predictor=pred, saved_feature_dir=str(feat_dir),
                         save_predictions=str(tmp_path / "pred.tsv"))
    ocrnn.eval(); opred.eval()
    with torch.no_grad():
        e, _ = ocrnn(torch.from_numpy(x))
        strong, _ = opred(e)
    ref = []
    for j in range(B):
        for lab, on, off in lo.post_process(strong[j].numpy(), median_window=5):
            ref.append((lab, round(on, 6), round(off, 6), f"clip{j}"))
    got = [(r.event_label, round(r.onset, 6), round(r.offset, 6), r.filename) for r in df.itertuples()]
    assert len(ref) > 0 and sorted(got) == sorted(ref)
    dumped = np.load(feat_dir / "0.npy")
    np.testing.assert_allclose(dumped, e.numpy(), atol=1e-4)
    assert (tmp_path / "pred.tsv").exists()
    assert crnn.training and pred.training  # restored
