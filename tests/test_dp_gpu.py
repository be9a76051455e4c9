"""Data parallelism on hardware with the means a one-GPU box offers: two FRESH child processes (started before they
touch the GPU) share cuda:0, exchange gradients through torch.distributed ("gloo": host-staged, so two ranks can sit on
one card) and run SEDTrainer steps on their clip shards (rank::2, per-rank dropout seeds, per-rank BatchNorm
statistics).  The summed gradient arena every rank ends up with must equal the sum of the two per-rank arenas computed in
THIS process (each rank emulated with world size 1) -- bitwise, because every kernel is bitwise repeatable and a
two-term fp32 sum is order-independent.  Covers the one-arena layout, the early / tail split of the exchange started
from inside the backward pass (plain step: first backward; mean-teacher step: second backward) and
broadcast_parameters.  The 1/world factor lives in the optimizer kernel (tested in test_parallel_cpu.py)."""
import os
import socket
import subprocess
import sys

import pytest
import torch

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    return str(port)


def test_two_rank_gradient_exchange_matches_per_rank_gradients(tmp_path):
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import dp_worker
    port = _free_port()
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0")
    procs = [subprocess.Popen([sys.executable, os.path.join(ROOT, "tests", "dp_worker.py"), str(r), "2", port, str(tmp_path)],
                              env=env, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True) for r in range(2)]
    logs = []
    for p in procs:
        try:
            out, _ = p.communicate(timeout=420)
        except subprocess.TimeoutExpired:
            for q in procs:
                q.kill()
            raise
        logs.append(out)
    assert all(p.returncode == 0 for p in procs), "\n".join(logs)
    got = [torch.load(os.path.join(tmp_path, f"rank{r}.pt")) for r in range(2)]
    mine = [dp_worker.run_steps(r, 2, group_ready=False) for r in range(2)]
    for key in ("plain", "mt"):
        want = mine[0][key] + mine[1][key]
        assert float(want.abs().max()) > 0
        assert torch.equal(got[0][key], got[1][key]), key          # both ranks hold the same reduced arena
        assert torch.equal(got[0][key], want), (key, float((got[0][key] - want).abs().max()))
        # the per-rank gradients really differ (different clips, different dropout seeds)
        assert not torch.equal(mine[0][key], mine[1][key])
    tail = int(got[0]["tail_floats"])
    assert 0 < tail < got[0]["plain"].numel() // 50              # the late segment is the first two blocks: tiny
    for r in range(2):
        assert abs(float(got[r]["plain_loss"]) - float(mine[r]["plain_loss"])) == 0.0


def test_single_rank_rccl_group_runs_the_exchange_path(tmp_path):
    """SEDTrainer under a 1-rank "nccl" (= RCCL) process group with the exchange forced on: begin_early is issued from
    inside the last backward pass and finish waits for early + tail before the optimizer, on RCCL's own stream.  The
    all-reduce of one rank is the identity, so the arena must equal, bit for bit, the one a group-less process computes."""
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import dp_worker
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0")
    p = subprocess.Popen([sys.executable, os.path.join(ROOT, "tests", "dp_worker.py"), "0", "1", _free_port(), str(tmp_path),
                          "nccl"], env=env, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)
    try:
        out, _ = p.communicate(timeout=420)
    except subprocess.TimeoutExpired:
        p.kill()
        raise
    assert p.returncode == 0, out
    got = torch.load(os.path.join(tmp_path, "rank0.pt"))
    mine = dp_worker.run_steps(0, 1, group_ready=False)
    assert int(got["exchanges"]) == 4 and int(mine["exchanges"]) == 0     # early + tail, plain and mean-teacher step
    for key in ("plain", "mt"):
        assert float(mine[key].abs().max()) > 0
        assert torch.equal(got[key], mine[key]), (key, float((got[key] - mine[key]).abs().max()))
