"""Train step / EMA / schedules: the host-side mirror of one iteration of the reference's ``train_mt``.

  train_step            <- reference src/main_baseline.py:197-598 (one loop iteration, no ISP):
                           student forward on the synthetic and the real batch, optional EMA-teacher
                           forward on the noisy real batch, BCE strong/weak + MSE consistency, backward,
                           optimizer step, EMA update.
  update_ema_variables  <- src/main_baseline.py:91-105
  adjust_learning_rate  <- src/main_baseline.py:53-88
  ramps                 <- src/utilities/ramps.py:4-30

Everything between the waveform batch and the updated parameters runs in HIP kernels on the current
stream; the host only sequences launches (no ``.item()`` / device syncs inside the step, unlike the
reference's >= 6 syncs per iteration).  Data parallelism: one process per GPU; the gradients of CRNN, Predictor
and discriminator live in ONE flat arena (``parallel.GradArena``) whose all-reduce (RCCL through
``torch.distributed``, backend "nccl") is started from inside the last backward pass, as soon as every gradient
but the first two CNN blocks' has been enqueued, and overlaps the rest of it; the 1/world factor is folded into
the optimizer kernel.
"""
import numpy as np
import ctypes
import os

import torch

from . import _lib as L
from . import ops
from . import parallel


# ----------------------------------------------------------------------------- schedules
def exp_rampup(current, rampup_length):
    if rampup_length == 0:
        return 1.0
    current = np.clip(current, 0.0, rampup_length)
    phase = 1.0 - current / rampup_length
    return float(np.exp(-5.0 * phase * phase))


def sigmoid_rampdown(current, rampup_length):
    if rampup_length == 0:
        return 1.0
    current = np.clip(current, 0.0, rampup_length)
    phase = 1.0 - current / rampup_length
    return float(np.exp(-12.5 * phase * phase))


def adjust_learning_rate(optimizer, rampup_value, rampdown_value=1, optimizer_d=None, optimizer_crnn=None,
                         c_epoch=None, rampup_value_adv=None, max_learning_rate=0.0005):
    """Same schedule and argument order as the reference; ``optimizer`` objects need ``param_groups``
    (torch optimizers) or an ``lr`` attribute (FlatAdam / FlatSGD below)."""
    lr = rampup_value * rampdown_value * max_learning_rate
    if c_epoch is not None and c_epoch > 100:
        lr = lr * (0.5 ** (1 + ((c_epoch - 100) // 20)))

    def _set(opt, value):
        if hasattr(opt, "param_groups"):
            for g in opt.param_groups:
                g["lr"] = value
        else:
            opt.lr = value
    _set(optimizer, lr)
    if optimizer_d is not None:
        _set(optimizer_d, lr * 0.1)
    if optimizer_crnn is not None:
        _set(optimizer_crnn, lr * 0.1)
    return lr


@torch.no_grad()
def update_ema_variables(model, ema_model, alpha, global_step):
    """ema = ema*alpha' + model*(1-alpha'), alpha' = min(1 - 1/(step+1), alpha), over EVERY state entry
    (parameters, BatchNorm running statistics and the int64 num_batches_tracked counters), one kernel per
    flat arena.  (The reference implementation raises for a plain CRNN because of its "cnn." key quirk --
    DESIGN.md D8 -- this is the update it intends.)"""
    alpha = min(1 - 1 / (global_step + 1), alpha)
    ops.ema_update(ema_model.flat, model.flat, alpha)
    if getattr(model, "flat_buf", None) is not None:
        ops.ema_update(ema_model.flat_buf, model.flat_buf, alpha)
    if getattr(model, "nbt", None) is not None:
        ops.ema_update_i64(ema_model.nbt, model.nbt, alpha)


# ----------------------------------------------------------------------------- optimizers on flat arenas
def _ref_params(modules):
    """[(module, name, offset, numel, shape)] in the reference optimizer's parameter order
    (``list(crnn.parameters()) + list(predictor.parameters())``, src/main_baseline.py:865)"""
    out = []
    for m in modules:
        names = m.reference_param_names() if hasattr(m, "reference_param_names") else [n for n, _ in m.named_parameters()]
        for n in names:
            p = m.P(n)
            out.append((m, n, m._poff[n], p.numel(), tuple(p.shape)))
    return out


class FlatAdam:
    """torch.optim.Adam(lr, betas, eps, weight_decay) over the flat arenas of several modules.  ``state_dict`` /
    ``load_state_dict`` speak torch.optim.Adam's format (per-parameter ``step`` / ``exp_avg`` / ``exp_avg_sq`` in
    the reference's parameter order, CPU tensors), so the ``optimizer`` entry of a checkpoint loads on either side."""

    def __init__(self, modules, lr=0.001, betas=(0.9, 0.999), eps=1e-8, weight_decay=0.0):
        self.modules = list(modules)
        self.lr, self.betas, self.eps, self.weight_decay = lr, betas, eps, weight_decay
        self.step_count = 0
        self.m = [torch.zeros_like(m.flat) for m in self.modules]
        self.v = [torch.zeros_like(m.flat) for m in self.modules]

    def zero_grad(self, set_to_none=False):
        for m in self.modules:
            m.flat_grad.zero_()

    def step(self, grad_scale=1.0):
        self.step_count += 1
        for mod, m, v in zip(self.modules, self.m, self.v):
            ops.adam_step(mod.flat, mod.flat_grad, m, v, self.lr, self.step_count, self.betas, self.eps,
                          self.weight_decay, grad_scale)

    def state_dict(self):
        state = {}
        for i, (mod, _, off, n, shape) in enumerate(_ref_params(self.modules)):
            k = self.modules.index(mod)
            if self.step_count > 0:
                state[i] = {"step": torch.tensor(float(self.step_count)),
                            "exp_avg": self.m[k][off:off + n].view(shape).detach().cpu().clone(),
                            "exp_avg_sq": self.v[k][off:off + n].view(shape).detach().cpu().clone()}
        group = {"lr": self.lr, "betas": tuple(self.betas), "eps": self.eps, "weight_decay": self.weight_decay,
                 "amsgrad": False, "maximize": False, "foreach": None, "capturable": False, "differentiable": False,
                 "fused": None, "params": list(range(len(_ref_params(self.modules))))}
        return {"state": state, "param_groups": [group]}

    def load_state_dict(self, sd):
        if "param_groups" not in sd:          # round-1 files: {"step", "lr", "m", "v"}
            self.step_count, self.lr = sd["step"], sd["lr"]
            for dst, src in zip(self.m + self.v, sd["m"] + sd["v"]):
                dst.copy_(src)
            return
        g = sd["param_groups"][0]
        self.lr, self.betas, self.eps = g["lr"], tuple(g["betas"]), g["eps"]
        self.weight_decay = g.get("weight_decay", 0.0)
        params = _ref_params(self.modules)
        if len(g["params"]) != len(params):
            raise L.BsedError(f"optimizer state has {len(g['params'])} parameters, the modules have {len(params)}")
        steps = set()
        for idx, (mod, _, off, n, shape) in zip(g["params"], params):
            st = sd["state"].get(idx)
            k = self.modules.index(mod)
            if st is None:
                self.m[k][off:off + n].zero_(); self.v[k][off:off + n].zero_()
                continue
            self.m[k][off:off + n].copy_(torch.as_tensor(st["exp_avg"]).reshape(-1))
            self.v[k][off:off + n].copy_(torch.as_tensor(st["exp_avg_sq"]).reshape(-1))
            steps.add(int(st["step"]))
        if len(steps) > 1:
            raise L.BsedError(f"per-parameter step counts differ ({sorted(steps)}): one flat Adam step cannot hold them")
        self.step_count = steps.pop() if steps else 0


class FlatSGD:
    """torch.optim.SGD(lr, momentum, nesterov=True, weight_decay) (reference main_scmt_ada_weak.py:854-866);
    state_dict in torch.optim.SGD's format (``momentum_buffer`` per parameter)."""

    def __init__(self, modules, lr=0.001, momentum=0.9, weight_decay=1e-4, nesterov=True):
        self.modules = list(modules)
        self.lr, self.momentum, self.weight_decay, self.nesterov = lr, momentum, weight_decay, nesterov
        self.step_count = 0
        self.buf = [torch.zeros_like(m.flat) for m in self.modules]

    def zero_grad(self, set_to_none=False):
        for m in self.modules:
            m.flat_grad.zero_()

    def step(self, grad_scale=1.0):
        for mod, buf in zip(self.modules, self.buf):
            ops.sgd_step(mod.flat, mod.flat_grad, buf, self.lr, self.momentum, self.weight_decay,
                         self.step_count == 0, self.nesterov, grad_scale)
        self.step_count += 1

    def state_dict(self):
        state = {}
        params = _ref_params(self.modules)
        for i, (mod, _, off, n, shape) in enumerate(params):
            k = self.modules.index(mod)
            state[i] = {"momentum_buffer": (self.buf[k][off:off + n].view(shape).detach().cpu().clone()
                                            if self.step_count > 0 else None)}
        group = {"lr": self.lr, "momentum": self.momentum, "dampening": 0, "weight_decay": self.weight_decay,
                 "nesterov": self.nesterov, "maximize": False, "foreach": None, "differentiable": False, "fused": None,
                 "params": list(range(len(params)))}
        return {"state": state, "param_groups": [group]}

    def load_state_dict(self, sd):
        g = sd["param_groups"][0]
        self.lr, self.momentum, self.weight_decay = g["lr"], g["momentum"], g.get("weight_decay", 0.0)
        self.nesterov = g.get("nesterov", True)
        params = _ref_params(self.modules)
        seen = False
        for idx, (mod, _, off, n, shape) in zip(g["params"], params):
            st = sd["state"].get(idx) or {}
            k = self.modules.index(mod)
            mb = st.get("momentum_buffer")
            if mb is None:
                self.buf[k][off:off + n].zero_()
            else:
                self.buf[k][off:off + n].copy_(torch.as_tensor(mb).reshape(-1)); seen = True
        self.step_count = 1 if seen else 0      # only "first step or not" matters to the kernel


# ----------------------------------------------------------------------------- the train step
class SEDTrainer:
    """One object = the state of a training run on ONE GPU (rank).  ``train_step`` is one iteration of the
    reference's ``train_mt`` loop body."""

    def __init__(self, crnn, predictor, ema_crnn=None, ema_predictor=None, optimizer=None, frontend=None,
                 max_consistency_cost=1.0, ema_alpha=0.999, process_group=None, seed=2023, domain_loss=None,
                 optimizer_d=None):
        """domain_loss: a ``disc.ConditionalDomainAdversarialLoss`` (adversarial variant, reference
        src/main_scmt_ada_weak.py:312-339,527-528,568-574); optimizer_d steps its discriminator."""
        self.domain_loss, self.optimizer_d = domain_loss, optimizer_d
        if domain_loss is not None and optimizer_d is None:
            self.optimizer_d = FlatSGD([domain_loss.domain_discriminator], lr=0.0005 * 0.1)
        self.crnn, self.predictor = crnn, predictor
        self.ema_crnn, self.ema_predictor = ema_crnn, ema_predictor
        self.optimizer = optimizer or FlatAdam([crnn, predictor], lr=0.0005)
        self.frontend = frontend
        self.max_consistency_cost, self.ema_alpha = max_consistency_cost, ema_alpha
        self.global_step = 0
        self.seed = seed
        self.pg = process_group
        self._prefetched, self._feat_stream = {}, None   # train_step(..., next_waves=...): features one step ahead
        self._uploaded, self._copy_stream, self._slots = {}, None, {}   # host waveforms: uploads one step ahead
        # mean teacher: the EMA pair's forward on its own stream beside the student's passes (BSED_TEACHER_OVERLAP=0: inline)
        self.teacher_overlap = os.environ.get("BSED_TEACHER_OVERLAP", "1") != "0"
        self._teacher_stream = None
        self.step_priority = os.environ.get("BSED_STEP_PRIORITY", "1") != "0"
        self._step_stream = None
        # the packed weight copies a step needs, made in ONE launch at its start from the second step on
        # (ops.PackPlan; BSED_PACK_PLAN=0: one launch per weight at first use, the pre-plan behaviour)
        self._pack_plans = {} if os.environ.get("BSED_PACK_PLAN", "1") != "0" else None
        self.world = 1
        self.rank = 0
        if process_group is not None or (torch.distributed.is_available() and torch.distributed.is_initialized()):
            self.world = torch.distributed.get_world_size(process_group)
            self.rank = torch.distributed.get_rank(process_group)
        # one gradient arena for everything this rank differentiates: [CRNN (first blocks lead) | Predictor | D]
        disc = domain_loss.domain_discriminator if domain_loss is not None else None
        self.arena = parallel.GradArena([crnn, predictor, disc], tail_floats=crnn.tail_grad_floats(),
                                        group=process_group)

    def _plan(self, which):
        if self._pack_plans is None:
            return None
        return self._pack_plans.setdefault(which, ops.PackPlan())

    def broadcast_parameters(self, src=0):
        """identical initial weights / statistics on every rank"""
        if self.world == 1:
            return
        disc = self.domain_loss.domain_discriminator if self.domain_loss is not None else None
        for m in (self.crnn, self.predictor, self.ema_crnn, self.ema_predictor, disc):
            if m is None:
                continue
            bufs = [m.flat] + ([m.flat_buf] if getattr(m, "flat_buf", None) is not None else [])
            parallel.broadcast_flat(bufs, src, self.pg)
            if getattr(m, "nbt", None) is not None:
                parallel.broadcast_flat([m.nbt], src, self.pg)

    def _device_wave(self, wav):
        """A waveform batch as a device tensor.  Device tensors pass through.  A HOST tensor (pinned memory, if the
        upload is to overlap anything) that a previous ``train_step(next_waves=...)`` announced was uploaded on the copy
        stream one step ago: the current stream waits for that copy's event.  An unannounced host tensor is uploaded
        here, on the current stream, in front of its consumer (blocking upload: the PCIe time is then inside the step)."""
        if wav.is_cuda:
            return wav
        up = self._uploaded.pop(id(wav), None)
        if up is not None and up[0] is wav:
            cur = torch.cuda.current_stream()
            cur.wait_event(up[2])
            up[1].record_stream(cur)
            return up[1]
        return wav.to(self.crnn.flat.device, non_blocking=True)

    def _upload_ahead(self, waves, step):
        """Start the host -> device copies of the NEXT step's waveforms on the copy stream (the copy engine works beside
        this step's kernels: 226 MB per 256 clips = ~4 ms of PCIe at 57 GB/s).  Two device slots per
        shape alternate: the slot refilled now held the batch of two steps ago, whose transform was enqueued on the feature
        stream during the step before last."""
        host = [w for w in waves if w is not None and not w.is_cuda]
        if not host:
            return
        if self._copy_stream is None:
            self._copy_stream = torch.cuda.Stream()
        dev = self.crnn.flat.device
        cs = self._copy_stream
        cs.wait_stream(torch.cuda.current_stream())
        if self._feat_stream is not None:
            cs.wait_stream(self._feat_stream)
        with torch.cuda.stream(cs):
            for k, w in enumerate(host):
                key = (k, tuple(w.shape), w.dtype)
                slots = self._slots.setdefault(key, [None, None])
                i = step % 2
                if slots[i] is None:
                    slots[i] = torch.empty(w.shape, dtype=w.dtype, device=dev)
                slots[i].copy_(w, non_blocking=True)
                ev = torch.cuda.Event()
                ev.record()
                self._uploaded[id(w)] = (w, slots[i], ev, step)

    def _features(self, wav, noisy=False, step=None):
        pre = self._prefetched.pop(id(wav), None)
        if pre is not None and pre[0] is wav and pre[1] == (noisy, self.global_step):
            # computed on the feature stream during the previous step (train_step(..., next_waves=...)): same kernels,
            # same seed, same values
            torch.cuda.current_stream().wait_event(pre[3])
            for t in (pre[2] if isinstance(pre[2], tuple) else (pre[2],)):
                t.record_stream(torch.cuda.current_stream())
            return pre[2]
        dwav = self._device_wave(wav)
        T = self.frontend.num_frames(dwav.shape[1])
        return self.frontend.transform(dwav, max_frames=T, noisy=noisy,
                                       seed=parallel.rank_seed(self.seed, self.global_step if step is None else step, self.rank))

    def _drop_stale_inputs(self):
        """Entries announced for this step or an earlier one that nobody consumed (the caller passed other tensors: last
        batch of an epoch, a skipped batch, an exception between steps) would otherwise pin a waveform batch, its
        feature tensors and an event forever."""
        for table, tag in ((self._prefetched, lambda v: v[1][1]), (self._uploaded, lambda v: v[3])):
            for k in [k for k, v in table.items() if tag(v) <= self.global_step]:
                del table[k]

    def _prefetch_features(self, waves):
        """Enqueue the NEXT step's waveform -> dB-mel transforms on the feature stream.  Called from the CRNN's
        recurrence hook: the two GRU layers are latency-bound and occupy half the chip (one workgroup per 4 batch rows),
        the mel kernels run beside them.  waves: [(wav, noisy), ...]; the caller must leave the waveform tensors
        untouched until the next train_step has consumed them.  Host waveforms were put on the copy stream at the start
        of this step (_upload_ahead); the feature stream waits for their events."""
        if self._feat_stream is None:
            self._feat_stream = torch.cuda.Stream()
        main = torch.cuda.current_stream()
        self._feat_stream.wait_stream(main)
        with torch.cuda.stream(self._feat_stream):
            for wav, noisy in waves:
                feats = self._features(wav, noisy=noisy, step=self.global_step + 1)
                ev = torch.cuda.Event()
                ev.record()
                self._prefetched[id(wav)] = (wav, (noisy, self.global_step + 1), feats, ev)

    def _teacher_forward(self, x, step_seed):
        self.ema_crnn.train(); self.ema_predictor.train()
        self.ema_crnn.set_seed(step_seed * 4 + 2)
        enc_e, _ = self.ema_crnn.run_forward(x, save=False)
        strong_e, _, weak_e, _ = self.ema_predictor.run_forward(enc_e)
        return strong_e, weak_e

    def _all_reduce_grads(self):
        """wait for the early segment's all-reduce (started inside the last backward pass), exchange the tail"""
        self.arena.finish()

    def train_step(self, syn_x, syn_y, real_x=None, real_y_weak=None, real_x_ema=None, consistency_cost=None,
                   from_wave=False, next_waves=None):
        """One iteration (see ``_train_step`` for the arguments).  With the EMA teacher on its own stream the step itself
        runs on a HIGH-PRIORITY stream of the trainer: the teacher's forward (normal priority) then fills the CUs the
        student's passes leave idle instead of taking turns with them -- 17.23 -> 16.96 ms per mean-teacher step at
        B = 128 + 128, same kernels in the same per-stream order: same bits.  The caller's stream waits for the step
        (``BSED_STEP_PRIORITY=0``: run on the caller's stream; torch offers no priority BELOW the default one, so the
        side streams cannot be lowered instead).  Without a teacher stream the step stays on the caller's stream
        (measured: no gain for the plain step, 13.48-13.50 vs 13.52-13.53 ms)."""
        mt = self.ema_crnn is not None and real_x is not None
        if not (mt and self.teacher_overlap and self.step_priority):
            return self._train_step(syn_x, syn_y, real_x, real_y_weak, real_x_ema, consistency_cost, from_wave, next_waves)
        caller = torch.cuda.current_stream()
        if self._step_stream is None:
            self._step_stream = torch.cuda.Stream(priority=-1)
        self._step_stream.wait_stream(caller)
        with torch.cuda.stream(self._step_stream):
            out = self._train_step(syn_x, syn_y, real_x, real_y_weak, real_x_ema, consistency_cost, from_wave, next_waves)
        caller.wait_stream(self._step_stream)
        for v in out.values():
            if isinstance(v, torch.Tensor):
                v.record_stream(caller)
        return out

    def _train_step(self, syn_x, syn_y, real_x=None, real_y_weak=None, real_x_ema=None, consistency_cost=None,
                    from_wave=False, next_waves=None):
        """syn_x/real_x: (B,1,T,F) dB-mel batches, or (B,n) waveforms with ``from_wave=True`` (the mel stage then
        runs on the GPU inside the step).  syn_y: (B,T',C) strong targets; real_y_weak: (B,C).
        Mean teacher is active iff EMA models were given AND a real batch is passed.
        next_waves (with from_wave): ``(next_syn_wav, next_real_wav or None)`` -- the waveforms the NEXT call will be
        given.  Their mel transforms are enqueued on a second stream while this step's recurrences run (a two-deep
        input pipeline: every step still transforms one batch, one step ahead); the next call must pass the same
        tensor objects, unmodified.  Results are bit-identical to the unpipelined step.  The waveforms may be HOST
        tensors (pinned): the next step's are then uploaded on a copy stream at the start of this step, beside its
        kernels, into two alternating device slots (an unannounced host batch is uploaded in front of its transform).
        Returns a dict of DEVICE tensors with the per-term loss sums (no host sync)."""
        crnn, pred = self.crnn, self.predictor
        mt = self.ema_crnn is not None and real_x is not None
        if next_waves is not None and not from_wave:
            raise L.BsedError("train_step(next_waves=...) prefetches waveform features: it needs from_wave=True")
        if from_wave:
            if self.frontend is None:
                raise L.BsedError("train_step(from_wave=True) needs a MelFrontEnd")
            if next_waves is not None:
                nxt = [(next_waves[0], False)]
                if len(next_waves) > 1 and next_waves[1] is not None:
                    nxt.append((next_waves[1], mt and real_x_ema is None))
                crnn.rnn_hook = lambda: self._prefetch_features(nxt)   # one-shot: fires at the first recurrence
            syn_x = self._features(syn_x)
            if real_x is not None:
                if mt and real_x_ema is None:
                    real_x, real_x_ema = self._features(real_x, noisy=True)
                else:
                    real_x = self._features(real_x)
            self._drop_stale_inputs()
            if next_waves is not None:
                self._upload_ahead(next_waves, self.global_step + 1)     # host tensors only; device tensors: no-op
        step_seed = parallel.rank_seed(self.seed, self.global_step, self.rank)
        crnn.train(); pred.train()
        self.arena.zero_()
        adv = self.domain_loss is not None and real_x is not None
        if adv:
            self.domain_loss.domain_discriminator.train()
        out = {}
        B, Tp, C = syn_y.shape
        # (weights are constant until the optimizer step: packed / split copies of a weight tensor are made once and
        #  shared by the forward and backward passes of the step's batches -- ops.pack_cache)
        teacher = None
        if mt and self.teacher_overlap:
            # The EMA teacher's forward depends on nothing the student computes in this step: it is enqueued on its own
            # stream NOW and fills the chip beside the student's latency-bound kernels (recurrences, small launches);
            # the consistency loss of the real batch waits for its event.  Same kernels, same inputs: same bits.
            if self._teacher_stream is None:
                self._teacher_stream = torch.cuda.Stream()
            main = torch.cuda.current_stream()
            self._teacher_stream.wait_stream(main)       # features and last step's EMA update are ordered before it
            with torch.cuda.stream(self._teacher_stream), torch.no_grad(), ops.pack_cache(self._plan("teacher")):
                strong_e, weak_e = self._teacher_forward(real_x_ema if real_x_ema is not None else real_x, step_seed)
                ev = torch.cuda.Event()
                ev.record()
            teacher = (strong_e, weak_e, ev)
        with ops.pack_cache(self._plan("step")):
            # ---- student on the synthetic batch: strong + weak BCE
            crnn.set_seed(step_seed * 4 + 0)
            enc_s, ctx_s = crnn.run_forward(syn_x, save=True)
            saved_s = pred.run_forward(enc_s)
            y_weak_syn = ops.max_over_time(syn_y)
            dx, lp = pred.run_backward(enc_s, saved_s, y_strong=syn_y.contiguous(), y_weak=y_weak_syn)
            out["syn"] = lp
            dft = None
            if adv:
                # domain loss on the SAME encodings (the reference runs a second, numerically identical forward when
                # dropout is 0 -- SURVEY.md 8d); its feature gradients arrive through the gradient-reverse layer
                crnn.set_seed(step_seed * 4 + 1)
                enc_r, ctx_r = crnn.run_forward(real_x, save=True)
                out["domain"] = self.domain_loss(None, enc_s, None, enc_r)
                dfs, dft = self.domain_loss.backward_features()
                ops.axpy(dx, dfs)
            # the gradient exchange starts inside the LAST backward pass of the step
            last_is_syn = real_x is None or not (mt or dft is not None)
            crnn.run_backward(ctx_s, dx, on_early_grads=self.arena.begin_early if last_is_syn else None)
            del ctx_s
            # ---- student on the real batch (+ EMA teacher on its noisy twin)
            if real_x is not None:
                if not adv:
                    crnn.set_seed(step_seed * 4 + 1)
                    enc_r, ctx_r = crnn.run_forward(real_x, save=mt)
                saved_r = pred.run_forward(enc_r)
                if mt:
                    w = self.max_consistency_cost if consistency_cost is None else consistency_cost
                    if teacher is not None:
                        strong_e, weak_e, ev = teacher
                        torch.cuda.current_stream().wait_event(ev)
                        strong_e.record_stream(torch.cuda.current_stream())
                        weak_e.record_stream(torch.cuda.current_stream())
                    else:
                        with torch.no_grad():
                            strong_e, weak_e = self._teacher_forward(real_x_ema if real_x_ema is not None else real_x,
                                                                     step_seed)
                    dx, lp = pred.run_backward(enc_r, saved_r, y_weak=real_y_weak.contiguous(), ema_strong=strong_e,
                                               ema_weak=weak_e, w_cons_s=w, w_cons_w=w)
                    if dft is not None:
                        ops.axpy(dx, dft)
                    crnn.run_backward(ctx_r, dx, on_early_grads=self.arena.begin_early)
                    out["real"] = lp
                    del ctx_r
                elif dft is not None:
                    crnn.run_backward(ctx_r, dft.contiguous(), on_early_grads=self.arena.begin_early)
                    del ctx_r
        # ---- data-parallel gradient exchange + update
        self._all_reduce_grads()
        self.optimizer.step(grad_scale=1.0 / self.world)
        if adv:
            self.optimizer_d.step(grad_scale=1.0 / self.world)
        self.global_step += 1
        if mt:
            update_ema_variables(crnn, self.ema_crnn, self.ema_alpha, self.global_step)
            update_ema_variables(pred, self.ema_predictor, self.ema_alpha, self.global_step)
        out["shape"] = (B, Tp, C)
        return out

    # ------------------------------------------------------------------ HIP-graph replay of the plain step
    def capture_step(self, syn_x, syn_y, from_wave=False, warmup=3):
        """Capture the plain train step (synthetic batch only: BASELINE configs[2], the reference's ``train_mt`` without
        ``-mt``) of THIS shape in a HIP graph.  At the reference's batch of 24 (src/data/config.py:70) a step is ~95
        launches of 10-40 us each and the host needs 2.6 ms to enqueue what the GPU runs in < 2 ms: replaying a graph
        removes the host from the loop.  What changes from step to step lives in DEVICE memory: the dropout seed addend
        and the optimizer's step count (``bsed_set_step_state``), advanced by a node of the graph itself; inputs are
        copied into the graph's static tensors.  ``warmup`` eager steps run first (allocator, weight-pack plan, lazy
        tables).  Returns after the capture; then call ``replay_step(syn_x, syn_y)``.  Replayed steps are bit-identical
        to eager ones (tests/test_graph_step_gpu.py)."""
        if self.ema_crnn is not None or self.domain_loss is not None or self.world != 1:
            raise L.BsedError("capture_step covers the plain single-rank step (no EMA teacher, no discriminator, no "
                              "data-parallel group: their per-step host decisions are not graph nodes yet)")
        dev = self.crnn.flat.device
        self._g_x = syn_x.clone()
        self._g_y = syn_y.clone()
        self._g_from_wave = from_wave
        for _ in range(warmup):
            self._train_step(self._g_x, self._g_y, from_wave=from_wave)
        # device-resident step state: zero addends now, the baked scalars are those of THIS step
        self._g_state = torch.zeros(4, device=dev, dtype=torch.int64)     # [0] seed addend (uint64), [1] step addend (int32)
        L.check(L.lib().bsed_set_step_state(ctypes.c_void_p(self._g_state[0:1].data_ptr()),
                                            ctypes.c_void_p(self._g_state[1:2].data_ptr())), "bsed_set_step_state")
        self._g_base_step = self.global_step
        torch.cuda.synchronize()
        self._graph = torch.cuda.CUDAGraph()
        seed_inc = (parallel.rank_seed(self.seed, 1, self.rank) - parallel.rank_seed(self.seed, 0, self.rank)) * 4
        with torch.cuda.graph(self._graph):
            out = self._train_step(self._g_x, self._g_y, from_wave=from_wave)
            L.check(L.lib().bsed_step_state_advance(ctypes.c_void_p(self._g_state[0:1].data_ptr()),
                                                    ctypes.c_void_p(self._g_state[1:2].data_ptr()),
                                                    ctypes.c_uint64(seed_inc), ctypes.c_int(1), L.stream()),
                    "bsed_step_state_advance")
        self._g_out = out
        # the capture ran no kernel: undo its host-side bookkeeping (the first replay IS that step)
        self.global_step = self._g_base_step
        self.optimizer.step_count -= 1
        return self

    def replay_step(self, syn_x, syn_y):
        """one captured step on new inputs (same shapes and dtypes as at capture); returns the same dict of device
        tensors as ``train_step`` (overwritten by the next replay)"""
        self._g_x.copy_(syn_x, non_blocking=True)
        self._g_y.copy_(syn_y, non_blocking=True)
        self._graph.replay()
        self.global_step += 1
        self.optimizer.step_count += 1
        return self._g_out

    def release_graph(self):
        """back to eager steps: the library stops reading the device-resident step state"""
        L.check(L.lib().bsed_set_step_state(None, None), "bsed_set_step_state")
        self._graph = None

    # ------------------------------------------------------------------ ISP (shift-consistency) iteration
    def train_step_isp(self, syn_x, syn_y, real_x, real_y_weak, real_x_ema, shift_frames, shift_bins,
                       consistency_cost=None, pooling_time_ratio=4):
        """One iteration of ``train_mt`` with ``-mt -ISP`` (reference src/main_baseline.py:229-277,337-420,431-529):
        on top of the mean-teacher step, time-rolled and frequency-rolled views of the synthetic and the real batch go
        through the student (4 extra forward/backward passes) and of the noisy real batch through the teacher (2 extra
        forwards).  shift_frames[k] (a multiple of pooling_time_ratio) / shift_bins[k] are the per-sample rolls the
        reference draws with random.randint(-64,64)*4 / random.randint(-4,4); the first half of the real batch is the
        weakly labelled half (its weak targets enter the frequency-shift class loss).  dB-mel inputs (B,1,T,F)."""
        if self.ema_crnn is None:
            raise L.BsedError("ISP needs the EMA teacher (the reference's consistency_cost only exists with -mt)")
        crnn, pred, ema_c, ema_p = self.crnn, self.predictor, self.ema_crnn, self.ema_predictor
        cc = self.max_consistency_cost if consistency_cost is None else consistency_cost
        dev = syn_x.device
        B, Tp, C = syn_y.shape
        T, F = syn_x.shape[2], syn_x.shape[3]
        half = real_y_weak.shape[0] // 2
        sh = torch.as_tensor(list(shift_frames), dtype=torch.int32, device=dev)
        sf = torch.as_tensor(list(shift_bins), dtype=torch.int32, device=dev)
        sp = torch.as_tensor([int(v / pooling_time_ratio) for v in shift_frames], dtype=torch.int32, device=dev)
        step_seed = parallel.rank_seed(self.seed, self.global_step, self.rank)
        crnn.train(); pred.train(); ema_c.train(); ema_p.train()
        self.arena.zero_()
        syn_x, real_x, real_x_ema = syn_x.contiguous(), real_x.contiguous(), real_x_ema.contiguous()
        syn_y = syn_y.contiguous()
        y_weak_syn = ops.max_over_time(syn_y)
        n_s, n_w = B * Tp * C, B * C
        out = {"shape": (B, Tp, C)}

        def fwd(x, slot):
            crnn.set_seed(step_seed * 16 + slot)
            enc, ctx = crnn.run_forward(x, save=True)
            return enc, pred.run_forward(enc), ctx

        with ops.pack_cache(self._plan("isp")):   # six student and three teacher passes on unchanged weights: one pack per weight
            # base passes (identical to the mean-teacher step)
            enc_s, sv_s, ctx_s = fwd(syn_x, 0)
            enc_r, sv_r, ctx_r = fwd(real_x, 1)
            with torch.no_grad():
                def teacher(x, slot):
                    ema_c.set_seed(step_seed * 16 + slot)
                    e, _ = ema_c.run_forward(x, save=False)
                    st, _, wk, _ = ema_p.run_forward(e)
                    return st, wk
                strong_e, weak_e = teacher(real_x_ema, 8)
                strong_e_sh, _ = teacher(ops.roll(real_x_ema, B, T, F, sh=sh), 9)
                strong_e_fs, _ = teacher(ops.roll(real_x_ema, B, T, F, sw=sf), 10)
            strong_r_roll = ops.roll(sv_r[0], B, Tp, C, sh=sp)      # detached by construction
            strong_s_roll = ops.roll(sv_s[0], B, Tp, C, sh=sp)
            y_s_roll = ops.roll(syn_y, B, Tp, C, sh=sp)
            dx, out["syn"] = pred.run_backward(enc_s, sv_s, y_strong=syn_y, y_weak=y_weak_syn)
            crnn.run_backward(ctx_s, dx)
            dx, out["real"] = pred.run_backward(enc_r, sv_r, y_weak=real_y_weak.contiguous(), ema_strong=strong_e,
                                                ema_weak=weak_e, w_cons_s=cc, w_cons_w=cc)
            crnn.run_backward(ctx_r, dx)
            del ctx_s, ctx_r
            # real, time shift: 1/2 cc MSE vs teacher(shifted) + cc/2 MSE vs the rolled (detached) base prediction
            enc, sv, ctx = fwd(ops.roll(real_x, B, T, F, sh=sh), 2)
            dx, out["real_shift"] = pred.run_backward(enc, sv, ema_strong=strong_e_sh, w_cons_s=0.5 * cc,
                                                      ema_strong2=strong_r_roll, w_cons_s2=0.5 * cc)
            crnn.run_backward(ctx, dx)
            # real, frequency shift: 1/2 cc MSE vs teacher(freq-shifted); weak BCE on the weakly labelled half only
            enc, sv, ctx = fwd(ops.roll(real_x, B, T, F, sw=sf), 3)
            parts, lps = [], []
            for lo, hi, yw in ((0, half, real_y_weak[:half].contiguous()), (half, B, None)):
                if hi <= lo:
                    continue
                d, lp = pred.run_backward(enc[lo:hi], tuple(t[lo:hi] for t in sv), y_weak=yw, ema_strong=strong_e_fs[lo:hi],
                                          w_cons_s=0.5 * cc, n_strong=n_s, n_weak=max(half, 1) * C)
                parts.append(d); lps.append(lp)
            out["real_fshift_weak_half"], out["real_fshift_rest"] = lps[0], lps[-1]
            crnn.run_backward(ctx, torch.cat(parts, 0))
            # synthetic, time shift: strong BCE vs the rolled target + cc/2 MSE vs the rolled (detached) base prediction
            enc, sv, ctx = fwd(ops.roll(syn_x, B, T, F, sh=sh), 4)
            dx, out["syn_shift"] = pred.run_backward(enc, sv, y_strong=y_s_roll, ema_strong=strong_s_roll, w_cons_s=0.5 * cc)
            crnn.run_backward(ctx, dx)
            # synthetic, frequency shift: strong + weak BCE vs the unshifted targets
            enc, sv, ctx = fwd(ops.roll(syn_x, B, T, F, sw=sf), 5)
            dx, out["syn_fshift"] = pred.run_backward(enc, sv, y_strong=syn_y, y_weak=y_weak_syn)
            crnn.run_backward(ctx, dx, on_early_grads=self.arena.begin_early)
        del ctx
        self._all_reduce_grads()
        self.optimizer.step(grad_scale=1.0 / self.world)
        self.global_step += 1
        update_ema_variables(crnn, ema_c, self.ema_alpha, self.global_step)
        update_ema_variables(pred, ema_p, self.ema_alpha, self.global_step)
        out["isp"] = (cc, half)
        return out

    @staticmethod
    def isp_loss_value(out):
        """scalar the reference would log for an ISP iteration (host sync)"""
        B, Tp, C = out["shape"]
        cc, half = out["isp"]
        n_s, n_w = B * Tp * C, B * C
        g = {k: v.double().sum(0).cpu() for k, v in out.items() if isinstance(v, torch.Tensor)}
        loss = g["syn"][0] / n_s + g["syn"][1] / n_w                                   # strong + weak class (syn)
        loss += g["real"][1] / n_w + cc * (g["real"][2] / n_s + g["real"][3] / n_w)     # weak class (real) + consistency
        loss += g["syn_fshift"][1] / n_w + g["real_fshift_weak_half"][1] / (max(half, 1) * C)  # weak freq-shift class
        loss += g["syn_shift"][0] / n_s                                                 # strong shift class
        loss += g["syn_fshift"][0] / n_s                                                # strong freq-shift class
        loss += 0.5 * cc * (g["syn_shift"][2] / n_s + g["real_shift"][4] / n_s)        # consistency_loss_shift
        fs = g["real_fshift_weak_half"][2] + (g["real_fshift_rest"][2] if half < B else 0.0)
        loss += 0.5 * cc * (g["real_shift"][2] / n_s + fs / n_s)                         # 1/2 (strong shift + freq shift vs EMA)
        return float(loss)

    @staticmethod
    def loss_value(out, consistency_cost=1.0):
        """Host-side assembly of the scalar the reference logs (syncs: call it outside the timed loop)."""
        B, Tp, C = out["shape"]
        s = out["syn"].double().sum(0).cpu()
        loss = float(s[0] / (B * Tp * C) + s[1] / (B * C))
        if "real" in out:
            r = out["real"].double().sum(0).cpu()
            loss += float(r[1] / (B * C) + consistency_cost * (r[2] / (B * Tp * C) + r[3] / (B * C)))
        if "domain" in out:
            loss += float(out["domain"])
        return loss
