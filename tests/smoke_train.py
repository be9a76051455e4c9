"""One tiny train step of the hot path on cuda:0, checked against the CPU oracle (used by
__graft_entry__.smoke(); test infrastructure: lives under tests/, not in the product package, because it imports
the oracle as the checker)."""
import numpy as np
import torch


def run():
    from oracle import crnn_oracle as co
    from oracle import seeded
    from bsed_amd.engine import FlatAdam, SEDTrainer
    from bsed_amd.models import CRNN, Predictor
    seed, B, T = 3, 2, 64
    kw = dict(co.CRNN_KWARGS)
    kw["dropout"] = 0.0
    ocrnn, opred = co.CRNN(**kw), co.Predictor(**co.PREDICTOR_KWARGS)
    seeded.load_seeded(ocrnn, seed)
    seeded.load_seeded(opred, seed + 1)
    x = seeded.db_like_input(seed + 2, B, T)
    y = seeded.strong_targets(seed + 3, B, T // 4)
    ocrnn.train(); opred.train()
    loss_ref, out = co.train_losses(ocrnn, opred, torch.from_numpy(x), torch.from_numpy(y))
    crnn, pred = CRNN(**kw), Predictor(**co.PREDICTOR_KWARGS)
    crnn.load_state_dict(ocrnn.state_dict())
    pred.load_state_dict(opred.state_dict())
    tr = SEDTrainer(crnn, pred, optimizer=FlatAdam([crnn, pred], lr=1e-3))
    before = crnn.flat.clone()
    res = tr.train_step(torch.from_numpy(x).cuda(), torch.from_numpy(y).cuda())
    loss = SEDTrainer.loss_value(res)
    assert abs(loss - float(loss_ref)) < 1e-4 * abs(loss), (loss, float(loss_ref))
    assert not torch.equal(before, crnn.flat), "optimizer did not move the parameters"
    crnn.eval(); pred.eval(); ocrnn.eval(); opred.eval()
    print("smoke ok: train-step loss %.6f (oracle %.6f)" % (loss, float(loss_ref)))
