"""diagnostic: which gradient tensors differ between two identical train steps (bitwise)"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import torch
from test_b256_gpu import _oracle_pair, _product_pair, B, T
from oracle import seeded
from bsed_amd.engine import FlatAdam, SEDTrainer
Bq = int(sys.argv[1]) if len(sys.argv) > 1 else B
x = torch.from_numpy(seeded.db_like_input(60, Bq, T)).cuda()
y = torch.from_numpy(seeded.strong_targets(61, Bq, T // 4)).cuda()
ocrnn, opred = _oracle_pair(62)
res = []
for rep in range(3):
    crnn, pred = _product_pair(ocrnn, opred, 0.5)
    tr = SEDTrainer(crnn, pred, optimizer=FlatAdam([crnn, pred], lr=1e-3), seed=7)
    junk = torch.empty((rep + 1) << 22, device="cuda")
    tr.train_step(x, y)
    res.append({n: p.grad.clone() for n, p in crnn.named_parameters()})
    del junk
for rep in (1, 2):
    bad = [(n, float((res[0][n] - res[rep][n]).abs().max()), float(res[0][n].abs().max())) for n in res[0] if not torch.equal(res[0][n], res[rep][n])]
    print("rep", rep, "differing tensors:", bad)
