"""N>1 logic on CPU: world_size-2 gloo processes exercise the same helpers the GPU trainer uses
(bsed_amd.parallel): strided clip sharding, flat-buffer sum all-reduce with the 1/world factor applied by the
optimizer, parameter broadcast, per-rank seeds, max-over-ranks timing."""
import os

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp


def _toy_module(specs):
    """a bsed_amd.models._FlatModule built on CPU storage (the GPU check of its constructor is what the real modules add)"""
    from bsed_amd.models import _FlatModule
    m = _FlatModule()
    m._build(specs, [], "cpu")
    return m


def _worker(rank, world, port, q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from bsed_amd import parallel
    torch.manual_seed(0)
    # a tiny model whose parameters live in one flat arena, like the real modules
    flat = torch.randn(37)
    if rank != 0:
        flat.add_(1.0)  # deliberately different before the broadcast
    parallel.broadcast_flat([flat])
    W = flat[:30].view(3, 10)
    b = flat[30:33]
    X = torch.arange(8 * 10, dtype=torch.float32).view(8, 10) / 80.0
    Y = torch.arange(8 * 3, dtype=torch.float32).view(8, 3) / 24.0
    xs, ys = parallel.shard_batch(X, rank, world), parallel.shard_batch(Y, rank, world)
    # per-rank gradient of the per-rank MEAN loss, summed over ranks, scaled by 1/world == full-batch gradient
    W_ = W.clone().requires_grad_(); b_ = b.clone().requires_grad_()
    ((xs @ W_.T + b_ - ys) ** 2).mean().backward()
    g = torch.cat([W_.grad.flatten(), b_.grad, torch.zeros(4)])
    parallel.all_reduce_flat([g])
    g_dp = g / world
    Wf = W.clone().requires_grad_(); bf = b.clone().requires_grad_()
    ((X @ Wf.T + bf - Y) ** 2).mean().backward()
    g_full = torch.cat([Wf.grad.flatten(), bf.grad, torch.zeros(4)])
    tmax = parallel.max_over_ranks(1.0 + rank, torch.device("cpu"))
    # GradArena: two flat-arena modules (the real _FlatModule class on CPU storage) share ONE gradient buffer; the
    # exchange runs as early segment (started "inside the backward pass") + tail segment
    mods = [_toy_module([("cnn.conv0.weight", (4, 3)), ("cnn.conv1.weight", (5,)), ("rnn.w", (6, 2))]),
            _toy_module([("dense.weight", (3, 3))])]
    arena = parallel.GradArena(mods, tail_floats=12)
    assert arena.flat.numel() == 12 + 5 + 12 + 9 and arena.tail.numel() == 12
    assert all(p.grad.data_ptr() >= arena.flat.data_ptr() for m in mods for p in m.parameters())
    for k, m in enumerate(mods):
        for j, p in enumerate(m.parameters()):
            p.grad.fill_(float(rank + 1) * (10 * k + j + 1))
    arena.begin_early()
    arena.finish()
    want = torch.cat([torch.full((12,), 3.0), torch.full((5,), 6.0), torch.full((12,), 9.0), torch.full((9,), 33.0)])
    arena_ok = bool(torch.equal(arena.flat, want)) and bool(torch.equal(mods[1].flat_grad, torch.full((9,), 33.0)))
    arena.zero_()
    arena_ok = arena_ok and float(mods[0].P("rnn.w").grad.abs().sum()) == 0.0
    q.put((rank, flat.numpy().copy(), float((g_dp - g_full).abs().max()), tmax,
           parallel.shard_indices(7, rank, world), parallel.rank_seed(2023, 5, rank), arena_ok))
    dist.barrier()
    dist.destroy_process_group()


def test_two_rank_data_parallel_gloo():
    world = 2
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 29500 + (os.getpid() % 2000)
    procs = [ctx.Process(target=_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = sorted((q.get(timeout=120) for _ in range(world)), key=lambda r: r[0])
    for p in procs:
        p.join(60)
        assert p.exitcode == 0
    (r0, flat0, e0, t0, idx0, s0, a0), (r1, flat1, e1, t1, idx1, s1, a1) = res
    assert a0 and a1                                      # one-arena exchange: every parameter's .grad summed over ranks
    np.testing.assert_array_equal(flat0, flat1)           # broadcast made the replicas identical
    assert e0 < 1e-6 and e1 < 1e-6                        # sum-all-reduce * 1/world == full-batch gradient
    assert t0 == t1 == 2.0                                # max over ranks
    assert idx0 == [0, 2, 4, 6] and idx1 == [1, 3, 5]     # clips r::world
    assert s0 != s1                                       # per-rank Philox seeds


def test_single_process_helpers_are_noops():
    from bsed_amd import parallel
    m = _toy_module([("a.weight", (2, 2)), ("b.weight", (3,))])
    m.flat_grad.copy_(torch.arange(7.0))
    arena = parallel.GradArena([m], tail_floats=4)
    assert torch.equal(arena.flat, torch.arange(7.0)) and torch.equal(m.P("b.weight").grad, torch.tensor([4.0, 5.0, 6.0]))
    arena.begin_early(); arena.finish()                   # world size 1: nothing to exchange
    assert torch.equal(arena.flat, torch.arange(7.0))
    g = torch.ones(5)
    assert parallel.all_reduce_flat([g]) == [] and torch.equal(g, torch.ones(5))
    parallel.broadcast_flat([g])
    assert parallel.max_over_ranks(3.5, torch.device("cpu")) == 3.5


def test_stale_arena_fails_loudly():
    """a second GradArena over the same module rebinds its gradients: the first one must refuse to work on dead storage"""
    from bsed_amd import parallel
    m = _toy_module([("a.weight", (2, 2)), ("b.weight", (3,))])
    first = parallel.GradArena([m], tail_floats=4)
    first.zero_()
    second = parallel.GradArena([m], tail_floats=4)
    second.zero_()
    with pytest.raises(RuntimeError, match="stale"):
        first.zero_()


def _single_rank_worker(port, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    import torch.distributed as dist
    from bsed_amd import parallel
    dist.init_process_group("gloo", rank=0, world_size=1)
    m = _toy_module([("a.weight", (2, 2)), ("b.weight", (3,))])
    m.flat_grad.copy_(torch.arange(7.0))
    plain = parallel.GradArena([m], tail_floats=4)
    plain.begin_early(); plain.finish()
    forced = parallel.GradArena([m], tail_floats=4, exchange_single_rank=True)
    forced.begin_early(); forced.finish()
    q.put((plain.exchanges, forced.exchanges, bool(torch.equal(forced.flat, torch.arange(7.0)))))
    dist.destroy_process_group()


def test_single_rank_group_can_be_made_to_exchange():
    """exchange_single_rank: the early + tail all-reduces run (identity) in a group of one rank -- what the GPU test uses
    to put begin_early / finish through RCCL on a one-GPU box"""
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    p = ctx.Process(target=_single_rank_worker, args=(31500 + (os.getpid() % 2000), q))
    p.start()
    plain, forced, same = q.get(timeout=120)
    p.join(60)
    assert p.exitcode == 0 and plain == 0 and forced == 2 and same
