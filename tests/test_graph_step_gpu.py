"""HIP-graph replay of the plain train step (engine.SEDTrainer.capture_step / replay_step): the reference's batch of 24
(/root/reference/src/data/config.py:70 batch_size = 12 -> 24 clips per step, src/main_baseline.py:737-740) is host-bound
in eager mode.  A replayed step must equal the eager step BIT FOR BIT: the per-step scalars a capture bakes (dropout seed,
Adam step count) are also read from device memory (bsed_set_step_state) and advanced by a node of the graph."""
import pytest
import torch

from oracle import crnn_oracle as co
from oracle import seeded
from test_crnn_gpu import _mine, _oracle

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("B,T,mode", [(4, 128, "bf16x3"), (24, 865, "bf16x3"), (6, 128, "bf16")])
def test_replayed_steps_equal_eager_steps_bitwise(B, T, mode):
    from bsed_amd.engine import FlatAdam, SEDTrainer
    xs = [torch.from_numpy(seeded.db_like_input(70 + k, B, T)).cuda() for k in range(4)]
    ys = [torch.from_numpy(seeded.strong_targets(80 + k, B, T // 4)).cuda() for k in range(4)]
    ocrnn, opred = _oracle(0.5, 9)
    res = {}
    for how in ("eager", "graph"):
        crnn, pred = _mine(0.5, ocrnn, opred, mode)
        tr = SEDTrainer(crnn, pred, optimizer=FlatAdam([crnn, pred], lr=1e-3), seed=11)
        losses = []
        if how == "eager":
            for _ in range(3):
                tr.train_step(xs[0], ys[0])
            for k in (1, 2, 3):
                losses.append(SEDTrainer.loss_value(tr.train_step(xs[k], ys[k])))
        else:
            tr.capture_step(xs[0], ys[0], warmup=3)
            try:
                for k in (1, 2, 3):
                    losses.append(SEDTrainer.loss_value(tr.replay_step(xs[k], ys[k])))
            finally:
                tr.release_graph()
        torch.cuda.synchronize()
        res[how] = (losses, crnn.flat.clone(), pred.flat.clone(), crnn.flat_buf.clone(), tr.global_step,
                    tr.optimizer.step_count)
    assert res["eager"][0] == res["graph"][0], (res["eager"][0], res["graph"][0])
    for i in (1, 2, 3):
        assert torch.equal(res["eager"][i], res["graph"][i]), i
    assert res["eager"][4:] == res["graph"][4:] == (6, 6)
