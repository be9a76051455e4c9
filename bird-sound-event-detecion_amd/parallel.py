"""Data-parallel plumbing (one process per GPU, torch.distributed; backend "nccl" is RCCL over xGMI on
ROCm, "gloo" in the CPU tests).  The reference has no distributed code: this is the MI355X-side design of
SURVEY.md section 8(e) -- shard clips by rank, keep BatchNorm statistics per rank, exchange ONE flat gradient
buffer per module per step, fold 1/world into the optimizer."""
import os

import torch
import torch.distributed as dist


def shard_indices(n_items, rank, world):
    """rank r takes items r, r+world, ... (``clips r::world``)"""
    return list(range(rank, n_items, world))


def shard_batch(batch, rank, world):
    """split the leading (clip) axis of a tensor / array across ranks, strided by rank"""
    return batch[rank::world]


class GradArena:
    """ONE contiguous fp32 gradient buffer for several flat-arena modules (CRNN + Predictor [+ discriminator]).

    Every module's ``flat_grad`` (and through it every parameter's ``.grad``) becomes a view of the arena, laid out as
    ``[first module | second | ...]``.  ``tail_floats`` leading floats -- the gradients of the first module's first
    CNN blocks, the ones the backward pass produces LAST -- form the ``tail`` segment, the rest the ``early`` segment:

        begin_early()  all-reduces ``early`` asynchronously as soon as its last gradient has been written (the caller
                       invokes it from the backward pass, before the first blocks' kernels are enqueued), so the
                       exchange overlaps the ~25 % of the backward pass that is still to run;
        finish()       all-reduces the few KB of ``tail`` and waits for both.

    Sums only: the 1/world factor is folded into the optimizer kernel.  With world size 1 both are no-ops (and the
    arena still makes zero_grad one memset).  SURVEY.md section 8(e); the reference has no distributed code."""

    def __init__(self, modules, tail_floats=0, group=None, exchange_single_rank=None):
        """exchange_single_rank: run the two all-reduces even in a group of ONE rank (identity on the data, but they
        execute on the backend's stream with the real dependencies) -- the one way a single-GPU box can put
        begin_early / finish through RCCL (tests/test_dp_gpu.py); default from BSED_DP_SINGLE_RANK_EXCHANGE."""
        self.modules = [m for m in modules if m is not None]
        self.group = group
        if exchange_single_rank is None:
            exchange_single_rank = os.environ.get("BSED_DP_SINGLE_RANK_EXCHANGE", "0") == "1"
        self.exchange_single_rank = bool(exchange_single_rank)
        sizes = [m.flat_grad.numel() for m in self.modules]
        dev = self.modules[0].flat_grad.device
        self.flat = torch.zeros(sum(sizes), device=dev, dtype=torch.float32)
        off = 0
        for m, n in zip(self.modules, sizes):
            view = self.flat[off:off + n]
            old = m.flat_grad.clone()
            m.flat_grad = view
            for _, p in m.named_parameters():
                p.grad = None                    # _attach_grads rebinds every .grad into the new storage ...
            m._attach_grads()
            view.copy_(old)                      # ... and clears an arena it finds unbound: put the values back
            off += n
        self.tail_floats = int(tail_floats)
        self.tail = self.flat[:self.tail_floats]
        self.early = self.flat[self.tail_floats:]
        self._work = None
        self.exchanges = 0          # all-reduces issued so far (evidence for the tests / the bench line)
        self.timing = False         # bench.py: record events around the waits of finish()
        self.wait_events = []       # [(before, after)] pairs on the stream that waited

    @property
    def world(self):
        if not dist.is_available() or not dist.is_initialized():
            return 1
        return dist.get_world_size(self.group)

    def _exchanging(self):
        if not dist.is_available() or not dist.is_initialized():
            return False
        return dist.get_world_size(self.group) > 1 or self.exchange_single_rank

    def _check_bound(self):
        """A later GradArena (a second SEDTrainer over the same modules, a trainer rebuilt after resume) rebinds the
        modules' flat_grad to ITS storage: this arena would then clear and all-reduce dead storage while the gradients
        accumulate elsewhere.  Fail loudly instead (two pointer compares per module, host side)."""
        for m in self.modules:
            lo, hi = self.flat.data_ptr(), self.flat.data_ptr() + 4 * self.flat.numel()
            if not (lo <= m.flat_grad.data_ptr() < hi):
                raise RuntimeError(f"{type(m).__name__}.flat_grad no longer lives in this GradArena (rebound by another "
                                   "arena or trainer): this arena is stale")

    def zero_(self):
        self._check_bound()
        self.flat.zero_()

    def begin_early(self):
        if not self._exchanging() or self._work is not None:
            return
        self._check_bound()
        # async_op: the collective runs on the backend's own stream after everything enqueued so far on the current
        # stream, concurrently with what the caller enqueues next (RCCL); gloo stages through the host instead
        self._work = dist.all_reduce(self.early, op=dist.ReduceOp.SUM, group=self.group, async_op=True)
        self.exchanges += 1

    def finish(self):
        if not self._exchanging():
            return
        if self._work is None:
            self.begin_early()
        works = [self._work]
        if self.tail_floats:
            works.append(dist.all_reduce(self.tail, op=dist.ReduceOp.SUM, group=self.group, async_op=True))
            self.exchanges += 1
        ev = None
        if self.timing and self.flat.is_cuda:
            ev = (torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True))
            ev[0].record()
        for w in works:
            w.wait()
        if ev is not None:
            ev[1].record()
            self.wait_events.append(ev)
        self._work = None


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
