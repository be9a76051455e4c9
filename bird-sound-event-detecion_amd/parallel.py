"""Data-parallel plumbing (one process per GPU, torch.distributed; backend "nccl" is RCCL over xGMI on
ROCm, "gloo" in the CPU tests).  The reference has no distributed code: this is the MI355X-side design of
SURVEY.md section 8(e) -- shard clips by rank, keep BatchNorm statistics per rank, exchange ONE flat gradient
buffer per module per step, fold 1/world into the optimizer."""
import torch
import torch.distributed as dist


def shard_indices(n_items, rank, world):
    """rank r takes items r, r+world, ... (``clips r::world``)"""
    return list(range(rank, n_items, world))


def shard_batch(batch, rank, world):
    """split the leading (clip) axis of a tensor / array across ranks, strided by rank"""
    return batch[rank::world]


def all_reduce_flat(buffers, group=None, async_op=True):
    """sum-all-reduce each flat buffer in place (no averaging: the optimizer kernel takes 1/world)."""
    if not dist.is_available() or not dist.is_initialized() or dist.get_world_size(group) == 1:
        return []
    works = [dist.all_reduce(b, op=dist.ReduceOp.SUM, group=group, async_op=async_op) for b in buffers]
    if async_op:
        for w in works:
            w.wait()
    return works


def broadcast_flat(buffers, src=0, group=None):
    if not dist.is_available() or not dist.is_initialized() or dist.get_world_size(group) == 1:
        return
    for b in buffers:
        dist.broadcast(b, src, group=group)


def max_over_ranks(value, device, group=None):
    """bench timing: the slowest rank defines the step time"""
    t = torch.tensor([float(value)], device=device, dtype=torch.float64)
    if dist.is_available() and dist.is_initialized() and dist.get_world_size(group) > 1:
        dist.all_reduce(t, op=dist.ReduceOp.MAX, group=group)
    return float(t)


def rank_seed(base_seed, step, rank, streams_per_step=64):
    """distinct Philox seeds per (step, rank): dropout / noise draws differ across ranks, repeat across runs"""
    return int(base_seed) * 1000003 + int(step) * streams_per_step + int(rank)
