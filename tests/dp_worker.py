"""One rank of the two-process data-parallel check (spawned by tests/test_dp_gpu.py; also usable by hand):

    python tests/dp_worker.py <rank> <world> <port> <outdir> [backend]

Both ranks share cuda:0 (backend "gloo": the collective goes through the host, which is what lets several ranks sit on
one card).  With world = 1 and backend "nccl" the single rank is made to exchange anyway (GradArena.exchange_single_rank):
begin_early / finish then execute on RCCL's stream -- all of the N > 1 path that one GPU can run.  Each rank takes clips rank::world of a seeded batch, runs one plain step and one mean-teacher step through
SEDTrainer (dropout 0.5, per-rank seeds) with lr = 0 optimizers, and saves the all-reduced gradient arena."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def build(seed, dropout=0.5):
    import torch
    from oracle import crnn_oracle as co
    from oracle import seeded
    from bsed_amd.models import CRNN, Predictor
    kw = dict(co.CRNN_KWARGS); kw["dropout"] = dropout
    crnn, pred = CRNN(**kw), Predictor(**co.PREDICTOR_KWARGS)
    for m, s in ((crnn, seed), (pred, seed + 1)):
        vals = seeded.seeded_state({k: tuple(v.shape) for k, v in m.state_dict().items()}, s)
        m.load_state_dict({k: torch.from_numpy(v) for k, v in vals.items()})
    return crnn, pred


def batch(B=8, T=128):
    import torch
    from oracle import seeded
    x = torch.from_numpy(seeded.db_like_input(91, B, T)).cuda()
    y = torch.from_numpy(seeded.strong_targets(92, B, T // 4)).cuda()
    xr = torch.from_numpy(seeded.db_like_input(93, B, T)).cuda()
    xe = torch.from_numpy(seeded.db_like_input(94, B, T)).cuda()
    yw = y.max(1)[0].contiguous()
    return x, y, xr, xe, yw


def run_steps(rank, world, group_ready, force_exchange=False):
    """-> dict of CPU tensors: summed gradient arenas of the two steps (+ losses)"""
    import torch
    from bsed_amd.engine import FlatSGD, SEDTrainer
    x, y, xr, xe, yw = batch()
    sh = slice(rank, None, world)
    out = {"exchanges": 0}
    # plain step
    crnn, pred = build(7)
    tr = SEDTrainer(crnn, pred, optimizer=FlatSGD([crnn, pred], lr=0.0, momentum=0.0, weight_decay=0.0), seed=11)
    if not group_ready:
        tr.rank, tr.world = rank, 1          # single-process emulation of one rank: same seeds, no exchange
    tr.arena.exchange_single_rank = force_exchange
    res = tr.train_step(x[sh].contiguous(), y[sh].contiguous())
    out["exchanges"] += tr.arena.exchanges
    out["plain"] = tr.arena.flat.detach().cpu().clone()
    out["plain_loss"] = torch.tensor(SEDTrainer.loss_value(res))
    # mean-teacher step (two backward passes; the exchange starts inside the second)
    crnn, pred = build(7)
    ema_c, ema_p = build(17)
    tr = SEDTrainer(crnn, pred, ema_c, ema_p, optimizer=FlatSGD([crnn, pred], lr=0.0, momentum=0.0, weight_decay=0.0), seed=11)
    if not group_ready:
        tr.rank, tr.world = rank, 1
    tr.arena.exchange_single_rank = force_exchange
    tr.train_step(x[sh].contiguous(), y[sh].contiguous(), xr[sh].contiguous(), yw[sh].contiguous(), xe[sh].contiguous())
    out["exchanges"] += tr.arena.exchanges
    out["mt"] = tr.arena.flat.detach().cpu().clone()
    out["tail_floats"] = torch.tensor(tr.arena.tail_floats)
    return out


def main():
    rank, world, port, outdir = int(sys.argv[1]), int(sys.argv[2]), sys.argv[3], sys.argv[4]
    backend = sys.argv[5] if len(sys.argv) > 5 else "gloo"
    os.environ.update(RANK=str(rank), WORLD_SIZE=str(world), MASTER_ADDR="127.0.0.1", MASTER_PORT=port)
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    import torch
    import torch.distributed as dist
    torch.cuda.set_device(0)
    if backend == "nccl":
        dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device("cuda", 0))
    else:
        dist.init_process_group(backend, rank=rank, world_size=world)
    from bsed_amd.engine import SEDTrainer
    # broadcast_parameters: rank 1 starts from different weights and must end up with rank 0's
    crnn, pred = build(7 if rank == 0 else 8)
    tr = SEDTrainer(crnn, pred)
    assert tr.world == world and tr.rank == rank
    tr.broadcast_parameters()
    ref_c, _ = build(7)
    assert torch.equal(crnn.flat, ref_c.flat) and torch.equal(crnn.flat_buf, ref_c.flat_buf)
    # a group of ONE rank on RCCL: the early / tail all-reduces are identities, but they run on RCCL's stream with the
    # step's real dependencies (begin_early from inside the last backward pass, finish before the optimizer)
    out = run_steps(rank, world, group_ready=True, force_exchange=(world == 1))
    torch.save(out, os.path.join(outdir, f"rank{rank}.pt"))
    dist.barrier()
    dist.destroy_process_group()
    print(f"rank {rank} done", flush=True)


if __name__ == "__main__":
    main()
