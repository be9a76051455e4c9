"""CRNN / Predictor with the reference's constructor signatures, state_dict keys and forward
contracts, executed by hand-written HIP kernels (libbsed.so).

    CRNN(n_in_channel, nclass, attention=False, activation="Relu", dropout=0, train_cnn=True,
         rnn_type='BGRU', n_RNN_cell=64, n_layers_RNN=1, dropout_recurrent=0, cnn_integration=False,
         learned_post=False, **cnn_kwargs)            reference src/models/CRNN_GRL.py:142-204
    Predictor(nclass, attention=False, n_RNN_cell=64)  reference src/models/CRNN_GRL.py:430-460

Differences that are deliberate:
  * parameters are views into ONE flat fp32 arena per module (``.flat`` / ``.flat_grad``) so the optimizer,
    the EMA update and the data-parallel all-reduce are single launches;
  * activations are NHWC on the device; the (B,1,T,F) input and the (B,T',256) output have the
    reference's layouts;
  * dropout masks come from a counter RNG (Philox) keyed on ``(seed, layer)``: forward and backward
    regenerate them instead of storing them.  Call ``set_seed(step)`` per step.
Only the hot-path configuration is built: GLU activation, 3x3/stride-1/pad-1 convs, BGRU with 128 cells
and 2 layers, 20 classes; anything else raises NotImplementedError (no silent fallback).
"""
import math
import re
from collections import OrderedDict

import numpy as np
import os

import torch
import torch.nn as nn

from . import _lib as L
from . import ops

BN_EPS = 1e-3
BN_MOMENTUM = 0.99


class _Holder(nn.Module):
    """Parameter container (no forward): gives the reference's dotted state_dict names."""


def _alloc_views(specs, device):
    """specs: [(name, shape)] -> (flat tensor, {name: view})"""
    total = sum(int(np.prod(s)) for _, s in specs)
    flat = torch.zeros(total, device=device, dtype=torch.float32)
    views, off = OrderedDict(), 0
    for name, shape in specs:
        n = int(np.prod(shape))
        views[name] = (off, flat[off:off + n].view(shape))
        off += n
    return flat, views


class _FlatModule(nn.Module):
    """nn.Module whose parameters / float buffers are views of flat arenas."""

    def _build(self, param_specs, buffer_specs, device):
        self.flat, pviews = _alloc_views(param_specs, device)
        self.flat_grad = torch.zeros_like(self.flat)
        self.flat_buf, bviews = _alloc_views(buffer_specs, device) if buffer_specs else (None, OrderedDict())
        self._poff = {k: v[0] for k, v in pviews.items()}
        self._boff = {k: v[0] for k, v in bviews.items()}
        for name, (off, view) in pviews.items():
            p = nn.Parameter(view)
            p.grad = self.flat_grad[off:off + view.numel()].view(view.shape)
            self._register(name, p, True)
        for name, (off, view) in bviews.items():
            self._register(name, view, False)

    def _register(self, dotted, tensor, is_param):
        parts = dotted.split(".")
        mod = self
        for p in parts[:-1]:
            if not hasattr(mod, p):
                mod.add_module(p, _Holder())
            mod = getattr(mod, p)
        if is_param:
            mod.register_parameter(parts[-1], tensor)
        else:
            mod.register_buffer(parts[-1], tensor)

    def P(self, dotted):
        mod = self
        for p in dotted.split("."):
            mod = getattr(mod, p)
        return mod

    def reference_param_names(self):
        """parameter names in the order ``reference_module.parameters()`` yields them: the index space of a
        torch.optim state_dict written by the reference's scripts (GRU tensors come per layer and direction as
        weight_ih, weight_hh, bias_ih, bias_hh there; the arena keeps both directions of a kind adjacent)."""
        own = [n for n, _ in self.named_parameters()]
        rnn = [n for n in own if ".rnn.weight_" in n or ".rnn.bias_" in n]
        if not rnn:
            return own
        def key(n):
            pfx, leaf = n.rsplit(".rnn.", 1)
            rev = leaf.endswith("_reverse")
            leaf = leaf[:-8] if rev else leaf
            kind, layer = leaf.rsplit("_l", 1)
            return (own.index(next(m for m in own if m.startswith(pfx + ".rnn."))), int(layer), rev,
                    ["weight_ih", "weight_hh", "bias_ih", "bias_hh"].index(kind))
        first = {}
        for n in rnn:
            first.setdefault(n.rsplit(".rnn.", 1)[0], own.index(n))
        out, done = [], set()
        for n in own:
            if n in rnn:
                pfx = n.rsplit(".rnn.", 1)[0]
                if pfx not in done:
                    done.add(pfx)
                    out += sorted((m for m in rnn if m.rsplit(".rnn.", 1)[0] == pfx), key=key)
            else:
                out.append(n)
        return out

    def G(self, dotted):
        return self.P(dotted).grad

    def zero_grad(self, set_to_none=False):
        self.flat_grad.zero_()
        self._attach_grads()

    def _attach_grads(self):
        """(Re)bind every parameter's .grad to its slice of flat_grad.  torch optimizers' zero_grad(set_to_none=
        True) drops the bindings; in that case the arena is cleared first, which is what the caller asked for."""
        params = list(self.named_parameters())
        if params and params[0][1].grad is None:
            self.flat_grad.zero_()
        for name, p in params:
            if p.grad is None or p.grad.data_ptr() != self.flat_grad.data_ptr() + 4 * self._poff[name]:
                off = self._poff[name]
                p.grad = self.flat_grad[off:off + p.numel()].view(p.shape)

    def _apply(self, fn, recurse=True):
        # parameters are views of a flat arena allocated on the GPU: moving them would break the views
        probe = fn(torch.zeros(1, device=self.flat.device))
        if probe.device != self.flat.device or probe.dtype != torch.float32:
            raise L.BsedError("bsed_amd modules live on the GPU in fp32; .to()/.cpu()/.half() are not supported")
        return self


def _check_cfg(cond, what):
    if not cond:
        raise NotImplementedError(f"bsed_amd builds only the reference hot-path configuration: {what}")


class CRNN(_FlatModule):
    _RUNS_RNN = True  # False: the GRU only owns state-dict entries (CRNN_pred)

    def __init__(self, n_in_channel, nclass, attention=False, activation="Relu", dropout=0, train_cnn=True,
                 rnn_type="BGRU", n_RNN_cell=64, n_layers_RNN=1, dropout_recurrent=0, cnn_integration=False,
                 learned_post=False, kernel_size=(3, 3, 3), padding=(1, 1, 1), stride=(1, 1, 1),
                 nb_filters=(64, 64, 64), pooling=((1, 4), (1, 4), (1, 4)), device="cuda"):
        super().__init__()
        L._require_gpu()
        _check_cfg(n_in_channel == 1 and not cnn_integration, "n_in_channel=1")
        _check_cfg(activation.lower() == "glu", 'activation="glu"')
        _check_cfg(rnn_type == "BGRU" and dropout_recurrent == 0, "BGRU without recurrent dropout")
        if self._RUNS_RNN:
            _check_cfg(n_RNN_cell == 128 and n_layers_RNN == 2, "128 GRU cells, 2 layers")
        self.nclass, self.n_layers = nclass, n_layers_RNN
        _check_cfg(all(k == 3 for k in kernel_size) and all(p == 1 for p in padding) and all(s == 1 for s in stride),
                   "3x3 / stride 1 / pad 1 convolutions")
        nb_filters = list(nb_filters)
        _check_cfg(nb_filters[0] in (16, 32) and all(f in (16, 32, 64, 128) for f in nb_filters)
                   and nb_filters[-1] == 128, "filters in {16,32,64,128}, last 128")
        pooling = [tuple(p) for p in pooling]
        _check_cfg(all(p in ((2, 2), (1, 2), (1, 1), (2, 1)) for p in pooling), "pooling windows of 1 or 2")
        self.n_in_channel, self.attention, self.rnn_type = n_in_channel, attention, rnn_type
        self.cnn_integration, self.train_cnn = cnn_integration, train_cnn
        self.nb_filters, self.pooling, self.dropout_p = nb_filters, pooling, float(dropout)
        self.n_hidden = n_RNN_cell
        self.seed = 0
        self.fused_glu_bwd = True  # False = the unfused 4-launch chain (kept as a cross-check in the tests)
        self.glu3 = os.environ.get("BSED_GLU3", "1") != "0"  # split-fp32 GLU kernels (csrc/glu3.hip)
        # GRU weight gradients of layer l on a side stream, beside the (latency-bound, half-chip) recurrence of layer
        # l-1: -0.26 ms per step at B = 256 (14.07 -> 13.81 ms), bitwise identical results (tests/test_fullsize_gpu.py).
        # The overlapped 1-tap weight gradients are stretched by the recurrence they share the chip with (0.076 ->
        # 0.167 ms on the hh shapes): ops.KernelTimer counts side-stream launches without timing them, and a rocprof
        # summary of the default configuration describes the overlap for those rows.  BSED_RNN_OVERLAP=0: one stream.
        self.overlap_rnn = os.environ.get("BSED_RNN_OVERLAP", "1") != "0"
        # first block without its conv output / gradient tensors in HBM (csrc/block0.hip); 0 = the four-kernel form
        # (conv0_fwd, glu16_fwd, glu16_bwd, conv0_wgrad), kept as the cross-check of the tests
        self.block0_fused = os.environ.get("BSED_BLOCK0_FUSED", "1") != "0"
        self._side_stream = None
        self.rnn_hook = None
        # eval-mode forwards (get_predictions, the CNN-only tagger): the packed weight copies of a forward in one launch
        self._eval_plan = ops.PackPlan()
        # "bf16x3" (default): the 3x3 conv forward / data-gradient contractions and the GRU projection GEMMs run on the
        # bf16 matrix cores with split-fp32 operands (csrc/igemm3.hip; measured 5.5e-6 on the logits of the reference
        # config, 18x inside the 1e-4 bar).  "fp32": exact fp32 matrix cores everywhere (9.6e-7 on the logits).
        import os as _os
        # "bf16" (BASELINE configs[1-2], SURVEY.md section 0 D4 / 8(d): the THROUGHPUT mode, not the parity mode): the CNN's
        # activations -- conv outputs, pooled outputs and every gradient tensor of the same shapes -- are bf16 in HBM and
        # its contractions are ONE bf16 MFMA per product; accumulation, bias, BatchNorm statistics and their backward map,
        # GLU gate math, master weights, weight gradients and the optimizer stay fp32.  The GRU and the head keep fp32
        # tensors and split-fp32 contractions (their tensors are 2 % of the step's bytes).  tests/test_bf16_mode_gpu.py
        # states its tolerance against the fp32 oracle.
        self.conv_mode = _os.environ.get("BSED_CONV_MODE", "bf16x3")
        if self.conv_mode not in ("fp32", "bf16x3", "bf16"):
            raise L.BsedError(f"BSED_CONV_MODE must be fp32, bf16x3 or bf16, got {self.conv_mode!r}")
        pspecs, bspecs = [], []
        cin = 1
        for i, co in enumerate(nb_filters):
            pspecs += [(f"cnn.conv{i}.weight", (co, cin, 3, 3)), (f"cnn.conv{i}.bias", (co,)),
                       (f"cnn.batchnorm{i}.weight", (co,)), (f"cnn.batchnorm{i}.bias", (co,)),
                       (f"cnn.glu{i}.linear.weight", (co, co)), (f"cnn.glu{i}.linear.bias", (co,))]
            bspecs += [(f"cnn.batchnorm{i}.running_mean", (co,)), (f"cnn.batchnorm{i}.running_var", (co,))]
            cin = co
        H = n_RNN_cell
        for l in range(n_layers_RNN):
            nin = nb_filters[-1] if l == 0 else 2 * H
            # forward and reverse tensors are adjacent so (768, nin) / (2,384,128) views cover both directions
            pspecs += [(f"rnn.rnn.weight_ih_l{l}", (3 * H, nin)), (f"rnn.rnn.weight_ih_l{l}_reverse", (3 * H, nin)),
                       (f"rnn.rnn.weight_hh_l{l}", (3 * H, H)), (f"rnn.rnn.weight_hh_l{l}_reverse", (3 * H, H)),
                       (f"rnn.rnn.bias_ih_l{l}", (3 * H,)), (f"rnn.rnn.bias_ih_l{l}_reverse", (3 * H,)),
                       (f"rnn.rnn.bias_hh_l{l}", (3 * H,)), (f"rnn.rnn.bias_hh_l{l}_reverse", (3 * H,))]
        ep, eb, extra_bn = self._extra_specs()
        self._build(pspecs + ep, bspecs + eb, device)
        self._init_order = self._init_roles()
        self.nbt = torch.zeros(len(nb_filters) + len(extra_bn), device=device, dtype=torch.int64)
        for i in range(len(nb_filters)):
            self.P(f"cnn.batchnorm{i}").register_buffer("num_batches_tracked", self.nbt[i])
        for j, name in enumerate(extra_bn):
            self.P(name).register_buffer("num_batches_tracked", self.nbt[len(nb_filters) + j])
        self.reset_parameters()

    def _extra_specs(self):
        """(parameter specs, buffer specs, names of extra BatchNorm modules) of a subclass"""
        return [], [], []

    def _gru_roles(self, prefix):
        return [(prefix, "gru")]

    def _gru_matrices(self, prefix):
        """the >= 2-D GRU tensors in nn.GRU.parameters() order (the order the reference's weights_init walks them in)"""
        return [f"{prefix}.rnn.weight_{kind}_l{l}{sfx}" for l in range(self.n_layers) for sfx in ("", "_reverse")
                for kind in ("ih", "hh")]

    def _init_roles(self):
        """[(module prefix, role)] in the order ``reference_model.apply(weights_init)`` visits the
        modules that own parameters (src/models/CNN.py:43-69, src/models/CRNN_GRL.py:144-173): what a tensor IS
        (conv / bn / linear / gru) is recorded here, not guessed from its name."""
        order = []
        for i in range(len(self.nb_filters)):
            order += [(f"cnn.conv{i}", "conv"), (f"cnn.batchnorm{i}", "bn"), (f"cnn.glu{i}.linear", "linear")]
        return order + self._gru_roles("rnn")

    # ------------------------------------------------------------------ init / state
    @torch.no_grad()
    def reset_parameters(self):
        """PyTorch default init (what the reference modules have before weights_init is applied)."""
        for name, p in self.named_parameters():
            if "batchnorm" in name or ".bn_" in name:
                p.fill_(1.0 if name.endswith("weight") else 0.0)
            elif name.startswith("rnn"):
                p.uniform_(-1 / math.sqrt(self.n_hidden), 1 / math.sqrt(self.n_hidden))
            elif name.endswith("weight"):
                fan_in = p[0].numel()
                p.uniform_(-1 / math.sqrt(fan_in), 1 / math.sqrt(fan_in))
            else:
                w = self.P(name[:-4] + "weight")
                fan_in = w[0].numel()
                p.uniform_(-1 / math.sqrt(fan_in), 1 / math.sqrt(fan_in))
        for name, b in self.named_buffers():
            if name.endswith("running_var"):
                b.fill_(1.0)
            else:
                b.zero_()

    def set_seed(self, seed):
        self.seed = int(seed)

    @property
    def _rnn_mode(self):
        return "fp32" if self.conv_mode == "fp32" else "bf16x3"

    @property
    def _mfma3(self):
        """contractions on the bf16 matrix cores (split-fp32 operands, or single bf16 products in the bf16 mode)"""
        return self.conv_mode in ("bf16x3", "bf16")

    @property
    def act_dtype(self):
        """storage type of the CNN's activation / gradient tensors"""
        return torch.bfloat16 if self.conv_mode == "bf16" else torch.float32

    def _check_bf16_mode(self):
        if self.conv_mode == "bf16" and not (ops.igemm3_nsplit() and self.glu3 and self.block0_fused and self.fused_glu_bwd
                                             and not any(c.__name__ == "CRNN_fpn" for c in type(self).__mro__)
                                             and self.nb_filters[0] == 16 and all(f >= 32 for f in self.nb_filters[1:])):
            raise L.BsedError("conv_mode='bf16' is built for the plain CRNN / CRNN_pred path (fused first block with 16 "
                              "filters, N-split conv kernel, bf16-core GLU kernels); FPN variant and A/B switches: use bf16x3")

    def load_state_dict(self, state_dict, strict=True):
        # reference checkpoints carry "cnn.conv0.weight"; its loaders rewrite to "cnn.cnn." (save_features.py:48-52)
        sd = OrderedDict((k.replace("cnn.cnn.", "cnn.", 1) if k.startswith("cnn.cnn.") else k, v)
                         for k, v in state_dict.items())
        own = nn.Module.state_dict(self)
        missing = [k for k in own if k not in sd]
        unexpected = [k for k in sd if k not in own]
        if strict and (missing or unexpected):
            raise RuntimeError(f"load_state_dict: missing {missing}, unexpected {unexpected}")
        with torch.no_grad():
            for k, v in sd.items():
                if k in own:
                    own[k].copy_(torch.as_tensor(v).to(own[k].device))
        return torch.nn.modules.module._IncompatibleKeys(missing, unexpected)

    # ------------------------------------------------------------------ forward
    def _rnn_views(self, l, prefix="rnn"):
        H = self.n_hidden
        nin = self.nb_filters[-1] if l == 0 else 2 * H
        o = self._poff
        base = f"{prefix}.rnn."
        w_ih = self.flat[o[f"{base}weight_ih_l{l}"]:o[f"{base}weight_ih_l{l}"] + 6 * H * nin]
        w_hh = self.flat[o[f"{base}weight_hh_l{l}"]:o[f"{base}weight_hh_l{l}"] + 6 * H * H]
        b_ih = self.flat[o[f"{base}bias_ih_l{l}"]:o[f"{base}bias_ih_l{l}"] + 6 * H]
        b_hh = self.flat[o[f"{base}bias_hh_l{l}"]:o[f"{base}bias_hh_l{l}"] + 6 * H]
        return nin, w_ih, w_hh, b_ih, b_hh

    def _rnn_grads(self, l, prefix="rnn"):
        H = self.n_hidden
        nin = self.nb_filters[-1] if l == 0 else 2 * H
        o, g = self._poff, self.flat_grad
        base = f"{prefix}.rnn."
        return (g[o[f"{base}weight_ih_l{l}"]:o[f"{base}weight_ih_l{l}"] + 6 * H * nin],
                g[o[f"{base}weight_hh_l{l}"]:o[f"{base}weight_hh_l{l}"] + 6 * H * H],
                g[o[f"{base}bias_ih_l{l}"]:o[f"{base}bias_ih_l{l}"] + 6 * H],
                g[o[f"{base}bias_hh_l{l}"]:o[f"{base}bias_hh_l{l}"] + 6 * H])

    # ------------------------------------------------------------------ building blocks
    def _block_forward(self, a, B, Hh, Ww, cin, co, pool, names, drop, rng_stream, nbt, train, first=False, bn_pre=None):
        """conv3x3 -> BatchNorm -> GLU -> Dropout -> AvgPool (reference src/models/CNN.py:46-67).  names = (conv, bn,
        glu-linear) parameter prefixes.  Returns (pooled, saved-for-backward dict)."""
        ph, pw = pool
        cw, cb = self.P(names[0] + ".weight"), self.P(names[0] + ".bias")
        taps, wsrc, s_tap = self._conv_taps(cw, Ww)
        if first and co == 16 and self.block0_fused and 1 < Ww <= 256 and B * Hh * Ww < (1 << 31):
            return self._block0_forward(a, B, Hh, Ww, pool, names, drop, rng_stream, nbt, train, bn_pre)
        if self.conv_mode == "bf16" and (first or a.dtype != torch.bfloat16):
            raise L.BsedError("conv_mode='bf16': this block has no bf16 path (first block must be the fused 16-filter one)")
        if first:
            y, stats = ops.conv0_fwd(a, cw, cb, B, Hh, Ww, co, want_stats=train)
        else:
            epi = ops.EPI_STATS if train else ops.EPI_PLAIN
            if self._mfma3 and cin % 32 == 0:
                w3 = ops.pack_weight3(wsrc, len(taps), cin, co, s_tap, 9, cin * 9)
                y, stats = ops.igemm3(a, w3, co, B, Hh, Ww, cin, taps, bias=cb, epilogue=epi)
            elif self._mfma3 and cin == 16 and ops.igemm3s_supported(Ww, cin):
                w3s = ops.pack_weight3s(cw, 9, co, 1, 9, cin * 9)
                y, stats = ops.igemm3s(a, w3s, co, B, Hh, Ww, ops.TAPS3x3, bias=cb, epilogue=epi)
            else:
                wpk = ops.pack_weight(wsrc, len(taps), cin, co, s_tap, 9, cin * 9)
                y, stats = ops.igemm(a, wpk, co, B, Hh, Ww, cin, taps=taps, bias=cb, epilogue=epi)
        bn = self.P(names[1])
        if train:
            mean, invstd, scale, shift = ops.bn_finalize(stats, co, float(B * Hh * Ww), BN_EPS, BN_MOMENTUM,
                                                         bn.weight, bn.bias, bn.running_mean, bn.running_var, nbt)
        else:
            mean = invstd = None
            scale, shift = bn_pre or ops.bn_eval(co, BN_EPS, bn.weight, bn.bias, bn.running_mean, bn.running_var)
        glu = self.P(names[2])
        if co == 16:
            # 4 FLOP/B: HBM-bound streaming kernel instead of the MFMA tile kernel (csrc/glu_small.hip)
            pooled = ops.glu16_fwd(y, scale, shift, glu.weight, glu.bias, B, Hh, Ww, (ph, pw), drop, rng_stream,
                                   self.seed)
        elif self._mfma3 and self.glu3 and ops.glu_fwd3_supported(Ww, co, (ph, pw)):
            pooled = ops.glu_fwd3(y, scale, shift, glu.weight, glu.bias, B, Hh, Ww, co, (ph, pw), drop, rng_stream,
                                  self.seed)
        else:
            wg = ops.pack_weight(glu.weight, 1, co, co, 0, 1, co)
            pooled, _ = ops.igemm(y, wg, co, B, Hh, Ww, co, bias=glu.bias, epilogue=ops.EPI_GLU_POOL,
                                  a_scale=scale, a_shift=shift, e_src=y, e_scale=scale, e_shift=shift,
                                  pool=(ph, pw), drop_p=drop, rng_stream=rng_stream, seed=self.seed)
        blk = dict(inp=a, y=y, mean=mean, invstd=invstd, scale=scale, shift=shift, H=Hh, W=Ww, cin=cin, co=co,
                   pool=(ph, pw), names=names, drop=drop, rng=rng_stream, first=first)
        return pooled, blk

    def _block0_forward(self, a, B, Hh, Ww, pool, names, drop, rng_stream, nbt, train, bn_pre=None):
        """the first block with its conv output recomputed where needed instead of stored (csrc/block0.hip): batch
        statistics from x alone, then conv + BN + GLU + dropout + pool in one pass x -> pooled"""
        cw, cb = self.P(names[0] + ".weight"), self.P(names[0] + ".bias")
        bn, glu = self.P(names[1]), self.P(names[2])
        xr64 = None
        if train:
            stats, xr64 = ops.block0_stats(a, cw, cb, B, Hh, Ww)
            mean, invstd, scale, shift = ops.bn_finalize(stats, 16, float(B * Hh * Ww), BN_EPS, BN_MOMENTUM,
                                                         bn.weight, bn.bias, bn.running_mean, bn.running_var, nbt)
        else:
            mean = invstd = None
            scale, shift = bn_pre or ops.bn_eval(16, BN_EPS, bn.weight, bn.bias, bn.running_mean, bn.running_var)
        pooled = ops.block0_fwd(a, cw, cb, scale, shift, glu.weight, glu.bias, B, Hh, Ww, pool, drop, rng_stream,
                                self.seed, out_dtype=self.act_dtype)
        blk = dict(inp=a, y=None, xr64=xr64, mean=mean, invstd=invstd, scale=scale, shift=shift, H=Hh, W=Ww, cin=1,
                   co=16, pool=pool, names=names, drop=drop, rng=rng_stream, first=True)
        return pooled, blk

    @staticmethod
    def _conv_taps(cw, Ww):
        """taps of a 3x3 / pad 1 convolution and how to find their weights.  On a width-1 map (the FPN levels) the six
        taps with dw != 0 only ever see zero padding: the convolution is exactly its centre column, a 3x1 stencil."""
        if Ww > 1:
            return ops.TAPS3x3, cw, 1
        return ((-1, 0), (0, 0), (1, 0)), cw.view(-1)[1:], 3   # weight[co][ci][kh][1]: element offset 1, tap stride 3

    def _gru_forward(self, seq, B, T, prefix, save):
        """2-layer BiGRU (reference src/models/RNN.py:7-16): input projections as one GEMM, then the recurrence"""
        layers = []
        for l in range(2):
            nin, w_ih, w_hh, b_ih, b_hh = self._rnn_views(l, prefix)
            if self._mfma3:
                w3 = ops.pack_weight3(w_ih, 1, nin, 768, 0, 1, nin)
                xp, _ = ops.igemm3(seq, w3, 768, 1, B * T, 1, nin, ((0, 0),), bias=b_ih)
            else:
                wpk = ops.pack_weight(w_ih, 1, nin, 768, 0, 1, nin)
                xp, _ = ops.igemm(seq, wpk, 768, 1, B * T, 1, nin, bias=b_ih)
            if l == 0 and self.rnn_hook is not None:
                # one-shot: independent work for the chip's idle half while the recurrences run (SEDTrainer enqueues
                # the next batch's mel transform on its feature stream here)
                hook, self.rnn_hook = self.rnn_hook, None
                hook()
            out, gates = ops.gru_fwd(xp.view(B, T, 768), w_hh, b_hh, B, T, save_gates=save, mode=self._rnn_mode)
            layers.append(dict(inp=seq, out=out, gates=gates))
            seq = out
        return seq, layers

    def _side(self):
        if self._side_stream is None:
            self._side_stream = torch.cuda.Stream()
        return self._side_stream

    def _join_side(self):
        """the current stream waits for the side stream's weight-gradient work (before anything reads the gradient arena)"""
        if self._side_stream is not None:
            torch.cuda.current_stream().wait_stream(self._side_stream)

    def _gru_param_grads(self, lay, l, prefix, dxp, dgh, pih, phh, B, T):
        """bias and weight gradients of GRU layer l from its recurrence outputs (no consumer inside the backward pass)"""
        nin, w_ih, w_hh, b_ih, b_hh = self._rnn_views(l, prefix)
        g_wih, g_whh, g_bih, g_bhh = self._rnn_grads(l, prefix)
        if pih is not None:  # the recurrence kernel already summed the bias gradients over time per batch row
            ops.colsum(pih, pih.shape[0], 768, 768, g_bih)
            ops.colsum(phh, phh.shape[0], 768, 768, g_bhh)
        else:
            ops.colsum(dxp, B * T, 768, 768, g_bih)
            ops.colsum(dgh, B * T, 768, 768, g_bhh)
        part, G, KP, NP = ops.wgrad(lay["inp"], dxp, 1, B * T, 1, nin, 768)
        ops.reduce_partials(part, G, 1, KP, NP, nin, 768, g_wih, 0, 1, nin)
        for dr in range(2):
            part, G, KP, NP = ops.wgrad(lay["out"], dgh, B, T, 1, 128, 384, taps=((-1 if dr == 0 else 1, 0),),
                                        in_pitch=256, dy_pitch=768, in_offset=dr * 128, dy_offset=dr * 384)
            ops.reduce_partials(part, G, 1, KP, NP, 128, 384, g_whh, 0, 1, 128, dst_offset=dr * 384 * 128)

    def _gru_backward(self, layers, d, B, T, prefix):
        """returns dL/d(input sequence) (B,T,nin of layer 0); parameter gradients accumulate into flat_grad.  The
        recurrence of a layer is latency-bound and occupies half the chip (one workgroup per 4 batch rows), and nothing
        downstream needs a layer's WEIGHT gradients: they are enqueued on a side stream, where they run beside the next
        layer's recurrence (and, for layer 0, beside the head of the CNN backward).  `_join_side` closes the fork."""
        for l in (1, 0):
            nin, w_ih, w_hh, b_ih, b_hh = self._rnn_views(l, prefix)
            lay = layers[l]
            dxp, dgh, pih, phh = ops.gru_bwd(d, lay["out"], lay["gates"], w_hh, B, T, mode=self._rnn_mode)
            if self.overlap_rnn:
                main, side = torch.cuda.current_stream(), self._side()
                side.wait_stream(main)                      # dxp / dgh (and the zeroed gradient arena) are ready
                with torch.cuda.stream(side), ops.deferred_reductions():   # the side stream's own queue: same kernels
                    self._gru_param_grads(lay, l, prefix, dxp, dgh, pih, phh, B, T)
                for t in (dxp, dgh, pih, phh, lay["inp"], lay["out"]):
                    if t is not None:
                        t.record_stream(side)               # the allocator must not recycle them under the side stream
            else:
                self._gru_param_grads(lay, l, prefix, dxp, dgh, pih, phh, B, T)
            if self._mfma3:
                w3 = ops.pack_weight3(w_ih, 1, 768, nin, 0, nin, 1)
                d, _ = ops.igemm3(dxp, w3, nin, 1, B * T, 1, 768, ((0, 0),))
            else:
                wpk = ops.pack_weight(w_ih, 1, 768, nin, 0, nin, 1)
                d, _ = ops.igemm(dxp, wpk, nin, 1, B * T, 1, 768)
            d = d.view(B, T, nin)
        return d

    def _block_backward(self, blk, dpool, B, seed, need_dgrad=True):
        """backward of one conv/BN/GLU/dropout/pool block; returns dL/d(block input) or None for the first block"""
        Hh, Ww, cin, co = blk["H"], blk["W"], blk["cin"], blk["co"]
        ph, pw = blk["pool"]
        y = blk["y"]
        conv_n, bn_n, glu_n = blk["names"]
        glu, bn = self.P(glu_n), self.P(bn_n)
        drop_b, rng = blk["drop"], blk["rng"]
        if blk["first"] and y is None:
            # fused first block: y recomputed from x, g consumed in registers (BatchNorm-backward sums, Gx = sum g x_tap);
            # conv0's weight gradient is assembled from Gx and the input's tap correlations (csrc/block0.hip)
            cw, cb = self.P(conv_n + ".weight"), self.P(conv_n + ".bias")
            pdw, pdb, st2, pgx, G = ops.block0_bwd(blk["inp"], cw, cb, blk["scale"], blk["shift"], glu.weight,
                                                   glu.bias, dpool.contiguous(), B, Hh, Ww, (ph, pw), drop_b, rng, seed)
            ops.reduce_partials(pdw, G, 1, 16, 16, 16, 16, glu.weight.grad, 0, 16, 1)
            ops.stats_to_grad(pdb, co, 0, glu.bias.grad)
            coef = ops.bn_bwd(st2, co, float(B * Hh * Ww), bn.weight, blk["mean"], blk["invstd"], bn.weight.grad,
                              bn.bias.grad, None, blk["inp"], apply=False)
            ops.block0_wgrad_finish(pgx, G, blk["xr64"], coef, blk["mean"], cw, cb, cw.grad)
            return None
        if co == 16:
            # one streaming pass: y, d_pooled -> g + partials of dW_glu, db_glu and the BN-backward sums
            g, pdw, pdb, st2, G = ops.glu16_bwd(y, blk["scale"], blk["shift"], glu.weight, glu.bias,
                                                dpool.contiguous(), B, Hh, Ww, (ph, pw), drop_b, rng, seed)
            ops.reduce_partials(pdw, G, 1, 16, 16, 16, 16, glu.weight.grad, 0, 16, 1)
            ops.stats_to_grad(pdb, co, 0, glu.bias.grad)
        elif co in (32, 64) and self.fused_glu_bwd and self._mfma3 and self.glu3:
            # all three contractions on the bf16 cores, operands fetched in MFMA register layout (csrc/glu3.hip)
            g, pdw, pdb, st2, G, slabs = ops.glu_bwd3(y, blk["scale"], blk["shift"], glu.weight, glu.bias,
                                                      dpool.contiguous(), B, Hh, Ww, co, (ph, pw), drop_b, rng, seed)
            ops.reduce_partials(pdw, G * slabs, 1, co, co, co, co, glu.weight.grad, 0, co, 1)
            ops.stats_to_grad(pdb, co, 0, glu.bias.grad)
        elif co == 128 and self.fused_glu_bwd and self._mfma3 and self.glu3:
            # lin recompute + g on the bf16 cores; d_lin goes through HBM to a 1-tap weight-gradient contraction
            g, dlin, pdb, st2, G = ops.glu_bwd3n(y, blk["scale"], blk["shift"], glu.weight, glu.bias,
                                                 dpool.contiguous(), B, Hh, Ww, co, (ph, pw), drop_b, rng, seed)
            ops.stats_to_grad(pdb, co, 0, glu.bias.grad)
            part, Gw, KP, NP = ops.wgrad(y, dlin, B, Hh, Ww, co, co, a_scale=blk["scale"], a_shift=blk["shift"])
            ops.reduce_partials(part, Gw, 1, KP, NP, co, co, glu.weight.grad, 0, 1, co)
            del dlin
        elif co in (32, 64, 128) and self.fused_glu_bwd:
            # three chained MFMA contractions per tile, y read once, g written once (csrc/glu_bwd.hip)
            wfwd = ops.pack_weight(glu.weight, 1, co, co, 0, 1, co)
            g, pdw, pdb, st2, G, slabs = ops.glu_bwd_fused(y, blk["scale"], blk["shift"], wfwd, glu.weight,
                                                           glu.bias, dpool.contiguous(), B, Hh, Ww, co, (ph, pw),
                                                           drop_b, rng, seed)
            ops.reduce_partials(pdw, G * slabs, 1, co, co, co, co, glu.weight.grad, 0, co, 1)
            ops.stats_to_grad(pdb, co, 0, glu.bias.grad)
        else:
            # (1) recompute lin, form d_lin and the gate-branch term
            wg = ops.pack_weight(glu.weight, 1, co, co, 0, 1, co)
            tt = torch.empty_like(y)
            dlin, st = ops.igemm(y, wg, co, B, Hh, Ww, co, bias=glu.bias, epilogue=ops.EPI_GLU_BWD,
                                 a_scale=blk["scale"], a_shift=blk["shift"], e_src=y, e_scale=blk["scale"],
                                 e_shift=blk["shift"], e_dpool=dpool, out2=tt, pool=(ph, pw), drop_p=drop_b,
                                 rng_stream=rng, seed=seed)
            ops.stats_to_grad(st, co, 0, glu.bias.grad)
            # (2) dW_glu = d_lin^T @ bn(y)
            part, G, KP, NP = ops.wgrad(y, dlin, B, Hh, Ww, co, co, a_scale=blk["scale"], a_shift=blk["shift"])
            ops.reduce_partials(part, G, 1, KP, NP, co, co, glu.weight.grad, 0, 1, co)
            # (3) g = d_lin @ W_glu + gate term  (gradient w.r.t. the BatchNorm output), with BN-backward sums
            wgT = ops.pack_weight(glu.weight, 1, co, co, 0, co, 1)
            g, st2 = ops.igemm(dlin, wgT, co, B, Hh, Ww, co, epilogue=ops.EPI_ADD_STATS2, out=tt, out2=tt, e_src=y)
        cw = self.P(conv_n + ".weight")
        # conv bias feeds a train-mode BatchNorm: its gradient is exactly zero, leave it
        if blk["first"]:
            # (4') the first block's d_y is only consumed by conv0's weight gradient: BatchNorm backward is applied
            # on load there, the largest tensor of the network is neither rewritten nor re-read
            coef = ops.bn_bwd(st2, co, float(B * Hh * Ww), bn.weight, blk["mean"], blk["invstd"], bn.weight.grad,
                              bn.bias.grad, g, y, apply=False)
            part, G = ops.conv0_wgrad(blk["inp"], g, B, Hh, Ww, co, y=y, coef=coef, mean=blk["mean"])
            ops.reduce_partials(part, G, 9, 1, co, 1, co, cw.grad, 1, 9, 9)
            return None
        taps, wsrc, s_tap = self._conv_taps(cw, Ww)
        if self._mfma3:
            # (4) BatchNorm backward applied on load inside the weight-gradient kernel (d_y = A g + B (y - mean) + C),
            # which also writes d_y once for the data gradient: the separate apply pass over g and y is gone
            coef = ops.bn_bwd(st2, co, float(B * Hh * Ww), bn.weight, blk["mean"], blk["invstd"], bn.weight.grad,
                              bn.bias.grad, g, y, apply=False)
            dy = torch.empty_like(g) if need_dgrad else None
            part, G, KP, NP = ops.wgrad(blk["inp"], g, B, Hh, Ww, cin, co, taps=taps, bn_y=y, bn_coef=coef,
                                        bn_mean=blk["mean"], dy_out=dy)
        else:
            # (4) BatchNorm backward -> d_y in place
            ops.bn_bwd(st2, co, float(B * Hh * Ww), bn.weight, blk["mean"], blk["invstd"], bn.weight.grad,
                       bn.bias.grad, g, y)
            dy = g
            part, G, KP, NP = ops.wgrad(blk["inp"], dy, B, Hh, Ww, cin, co, taps=taps)
        ops.reduce_partials(part, G, len(taps), KP, NP, cin, co, cw.grad, s_tap, 9, cin * 9,
                            dst_offset=0 if Ww > 1 else 1)
        if not need_dgrad:
            return None
        flipped = [(-a, -b) for a, b in taps]
        if self._mfma3 and co == 32 and cin <= 32 and ops.igemm3s_supported(Ww, co):
            # data gradient of a 32-channel layer: all taps' weights resident in LDS (csrc/igemm3.hip, igemm3s)
            wds = ops.pack_weight3s(cw, 9, cin, 1, cin * 9, 9, K=co)
            d_in, _ = ops.igemm3s(dy, wds, cin, B, Hh, Ww, flipped)
        elif self._mfma3:
            wd3 = ops.pack_weight3(wsrc, len(taps), co, cin, s_tap, cin * 9, 9)
            d_in, _ = ops.igemm3(dy, wd3, cin, B, Hh, Ww, co, flipped)
        else:
            wd = ops.pack_weight(wsrc, len(taps), co, cin, s_tap, cin * 9, 9)
            d_in, _ = ops.igemm(dy, wd, cin, B, Hh, Ww, co, taps=flipped)
        return d_in

    def _cnn_forward(self, x, ctx):
        """the seven conv/BN/GLU/dropout/pool blocks; returns (a (B,T',1,C), T')"""
        B, _, Hh, Ww = x.shape
        train = self.training
        drop = self.dropout_p if train else 0.0
        a, cin = x, 1
        bn_pre = None
        if not train and len(self.nb_filters) <= 16:
            # eval mode: the running-statistics scale / shift of all blocks in one launch
            bns = [self.P(f"cnn.batchnorm{i}") for i in range(len(self.nb_filters))]
            bn_pre = ops.bn_eval_batch([(co, bn.weight, bn.bias, bn.running_mean, bn.running_var)
                                        for co, bn in zip(self.nb_filters, bns)], BN_EPS)
        for i, co in enumerate(self.nb_filters):
            ph, pw = self.pooling[i]
            if Ww % pw or Ww < 2 and pw > 1:
                raise L.BsedError(f"block {i}: width {Ww} not divisible by the pooling window")
            names = (f"cnn.conv{i}", f"cnn.batchnorm{i}", f"cnn.glu{i}.linear")
            a, blk = self._block_forward(a, B, Hh, Ww, cin, co, (ph, pw), names, drop, 100 + i, self.nbt[i:i + 1],
                                         train, first=(i == 0), bn_pre=None if bn_pre is None else bn_pre[i])
            if ctx is not None:
                ctx["blocks"].append(blk)
            cin = co
            Hh, Ww = Hh // ph, Ww // pw
        if Ww != 1:
            raise L.BsedError(f"frequency axis must pool down to 1, got {Ww}")
        return a, Hh

    TAIL_BLOCKS = 2   # the first CNN blocks' gradients are the last the backward pass writes (parallel.GradArena)

    def tail_grad_floats(self):
        """floats of flat_grad that belong to the first TAIL_BLOCKS CNN blocks (they lead the arena)"""
        k = min(self.TAIL_BLOCKS, len(self.nb_filters))
        return self._poff[f"cnn.conv{k}.weight"] if k < len(self.nb_filters) else self._poff["rnn.rnn.weight_ih_l0"]

    def _cnn_backward(self, ctx, dpool, on_early_grads=None):
        for i in range(len(self.nb_filters) - 1, -1, -1):
            if i == min(self.TAIL_BLOCKS, len(self.nb_filters)) - 1:
                self._join_side()      # the GRU weight gradients (side stream) belong to the early segment
                if on_early_grads is not None:
                    ops.flush_reductions()   # ... and the queued reductions that finish them
                    on_early_grads()   # every gradient outside the first TAIL_BLOCKS blocks has been enqueued
            dpool = self._block_backward(ctx["blocks"][i], dpool, ctx["B"], ctx["seed"])

    def run_forward(self, x, save=True):
        """x: (B,1,T,F) fp32 GPU tensor.  Returns (enc (B,T',256), ctx for run_backward or None)."""
        if x.dim() != 4 or x.shape[1] != 1:
            raise L.BsedError(f"CRNN expects (B,1,T,F), got {tuple(x.shape)}")
        x = x.contiguous().float()
        B = x.shape[0]
        train = self.training
        drop = self.dropout_p if train else 0.0
        ctx = {"B": B, "blocks": [], "train": train, "seed": self.seed, "x": x} if save else None
        self._check_bf16_mode()
        with ops.pack_cache(None if train else self._eval_plan):
            a, T = self._cnn_forward(x, ctx)
            if a.dtype != torch.float32:
                a = a.float()      # bf16 mode: the (B, T', 128) encoding enters the GRU as fp32 (28 MB at B = 256)
            seq, layers = self._gru_forward(a.view(B, T, self.nb_filters[-1]), B, T, "rnn", save)
        enc = ops.dropout(seq, drop, 200, self.seed) if drop > 0 else seq
        if save:
            ctx.update(T=T, layers=layers, drop=drop)
        return enc, ctx

    # ------------------------------------------------------------------ backward
    def run_backward(self, ctx, d_enc, on_early_grads=None):
        """Accumulates parameter gradients into ``flat_grad``; returns nothing (the input needs no grad).
        on_early_grads: called once every gradient except those of the first TAIL_BLOCKS CNN blocks has been enqueued
        (the data-parallel trainer starts its gradient all-reduce there, overlapping the rest of the backward pass)."""
        B, T = ctx["B"], ctx["T"]
        d = d_enc.contiguous()
        # weight / bias gradient reductions are queued and go out in one launch per flush (ops.deferred_reductions)
        with ops.deferred_reductions():
            if ctx["drop"] > 0:
                d = ops.dropout(d, ctx["drop"], 200, ctx["seed"])
            d = self._gru_backward(ctx["layers"], d, B, T, "rnn")
            if not self.train_cnn:
                self._join_side()
                ops.flush_reductions()
                if on_early_grads is not None:
                    on_early_grads()
                return
            self._cnn_backward(ctx, d.view(B, T, 1, self.nb_filters[-1]).to(self.act_dtype), on_early_grads)

    def forward(self, x):
        if torch.is_grad_enabled() and self.training:
            enc = _CRNNFunction.apply(x, self, self.P("cnn.conv0.weight"))
        else:
            enc, _ = self.run_forward(x, save=False)
        return enc, enc


class CRNN_fpn(CRNN):
    """Drop-in for the reference's feature-pyramid variant (src/models/CRNN_GRL.py:293-389 with
    src/models/CNN_FPN.py:33-100; the ``-fpn`` flag of every training script).

    On top of the seven CNN blocks, two more pyramid levels REUSE one conv3x3 / BatchNorm / GLU (``cnn.cnn_fcn``,
    ``cnn.bn_fcn``, ``cnn.glu``), each followed by Dropout(0.5) -- fixed, not the constructor's dropout -- and
    AvgPool((2,1)): T, T/2 and T/4 frames.  Three 2-layer BiGRUs (``rnn``, ``rnn_2``, ``rnn_4``) run on the three
    levels; the coarse outputs are upsampled along time (bilinear, align_corners=True) and fused by two 1x1
    convolutions over concatenated channels (``conv1x1_2``, ``conv1x1_4``).  The reference hard-codes the upsample
    sizes (156 and 313 frames, i.e. 10 s at 32 kHz); here they follow the input length (T/2 and T), which is the same
    thing for the reference's shapes.  ``cnn.conv1x1`` exists in the reference module but is never called; it is kept
    because it owns state-dict entries (its gradient stays zero)."""

    FPN_DROPOUT = 0.5

    def _extra_specs(self):
        H = self.n_hidden
        ps = [("cnn.cnn_fcn.weight", (128, 128, 3, 3)), ("cnn.cnn_fcn.bias", (128,)),
              ("cnn.glu.linear.weight", (128, 128)), ("cnn.glu.linear.bias", (128,)),
              ("cnn.bn_fcn.weight", (128,)), ("cnn.bn_fcn.bias", (128,)),
              ("cnn.conv1x1.weight", (128, 256, 1, 1)), ("cnn.conv1x1.bias", (128,))]
        for pfx in ("rnn_2", "rnn_4"):
            for l in range(2):
                nin = self.nb_filters[-1] if l == 0 else 2 * H
                ps += [(f"{pfx}.rnn.weight_ih_l{l}", (3 * H, nin)), (f"{pfx}.rnn.weight_ih_l{l}_reverse", (3 * H, nin)),
                       (f"{pfx}.rnn.weight_hh_l{l}", (3 * H, H)), (f"{pfx}.rnn.weight_hh_l{l}_reverse", (3 * H, H)),
                       (f"{pfx}.rnn.bias_ih_l{l}", (3 * H,)), (f"{pfx}.rnn.bias_ih_l{l}_reverse", (3 * H,)),
                       (f"{pfx}.rnn.bias_hh_l{l}", (3 * H,)), (f"{pfx}.rnn.bias_hh_l{l}_reverse", (3 * H,))]
        ps += [("conv1x1_2.weight", (256, 512, 1, 1)), ("conv1x1_2.bias", (256,)),
               ("conv1x1_4.weight", (256, 512, 1, 1)), ("conv1x1_4.bias", (256,))]
        bs = [("cnn.bn_fcn.running_mean", (128,)), ("cnn.bn_fcn.running_var", (128,))]
        return ps, bs, ["cnn.bn_fcn"]

    def _init_roles(self):
        # reference module order: src/models/CNN_FPN.py:66-77 (cnn, cnn_fcn, glu, bn_fcn, conv1x1) then
        # src/models/CRNN_GRL.py:304-336 (rnn, rnn_2, rnn_4, conv1x1_2, conv1x1_4)
        order = []
        for i in range(len(self.nb_filters)):
            order += [(f"cnn.conv{i}", "conv"), (f"cnn.batchnorm{i}", "bn"), (f"cnn.glu{i}.linear", "linear")]
        order += [("cnn.cnn_fcn", "conv"), ("cnn.glu.linear", "linear"), ("cnn.bn_fcn", "bn"), ("cnn.conv1x1", "conv")]
        for pfx in ("rnn", "rnn_2", "rnn_4"):
            order += self._gru_roles(pfx)
        return order + [("conv1x1_2", "conv"), ("conv1x1_4", "conv")]

    _BASE_KEY = re.compile(r"^cnn\.(conv\d+|batchnorm\d+|glu\d+)\.")

    def state_dict(self, *args, **kwargs):
        """the reference's CNN_FPN keeps its seven base blocks in a Sequential named ``cnn`` and (unlike CNN) does not
        strip that level: its keys are ``cnn.cnn.conv0.weight`` ... next to ``cnn.cnn_fcn.weight`` (CNN_FPN.py:41-77)"""
        sd = super().state_dict(*args, **kwargs)
        return OrderedDict((("cnn." + k if self._BASE_KEY.match(k) else k), v) for k, v in sd.items())

    # 1x1 convolution over channels of a (B,T,512) sequence == GEMM with the (256,512) weight
    def _fuse(self, cat, name, B, T):
        w, b = self.P(name + ".weight"), self.P(name + ".bias")
        if self.conv_mode == "bf16x3":
            w3 = ops.pack_weight3(w, 1, 512, 256, 0, 1, 512)
            out, _ = ops.igemm3(cat, w3, 256, 1, B * T, 1, 512, ((0, 0),), bias=b)
        else:
            wpk = ops.pack_weight(w, 1, 512, 256, 0, 1, 512)
            out, _ = ops.igemm(cat, wpk, 256, 1, B * T, 1, 512, bias=b)
        return out.view(B, T, 256)

    def _fuse_backward(self, cat, d_out, name, B, T):
        """accumulates dW, db of the 1x1 convolution; returns dL/d(cat) (B,T,512)"""
        w, b = self.P(name + ".weight"), self.P(name + ".bias")
        ops.colsum(d_out, B * T, 256, 256, b.grad)
        part, G, KP, NP = ops.wgrad(cat, d_out, 1, B * T, 1, 512, 256)
        ops.reduce_partials(part, G, 1, KP, NP, 512, 256, w.grad, 0, 1, 512)
        if self.conv_mode == "bf16x3":
            w3 = ops.pack_weight3(w, 1, 256, 512, 0, 512, 1)
            d_cat, _ = ops.igemm3(d_out, w3, 512, 1, B * T, 1, 256, ((0, 0),))
        else:
            wpk = ops.pack_weight(w, 1, 256, 512, 0, 512, 1)
            d_cat, _ = ops.igemm(d_out, wpk, 512, 1, B * T, 1, 256)
        return d_cat.view(B, T, 512)

    def run_forward(self, x, save=True):
        if x.dim() != 4 or x.shape[1] != 1:
            raise L.BsedError(f"CRNN_fpn expects (B,1,T,F), got {tuple(x.shape)}")
        x = x.contiguous().float()
        B = x.shape[0]
        train = self.training
        drop = self.dropout_p if train else 0.0
        dropf = self.FPN_DROPOUT if train else 0.0
        ctx = {"B": B, "blocks": [], "train": train, "seed": self.seed, "x": x} if save else None
        a, T = self._cnn_forward(x, ctx)
        T2, T4 = T // 2, (T // 2) // 2
        if T4 < 1:
            raise L.BsedError(f"CRNN_fpn needs at least 4 CNN output frames, got {T}")
        C = self.nb_filters[-1]
        names = ("cnn.cnn_fcn", "cnn.bn_fcn", "cnn.glu.linear")
        nbt = self.nbt[len(self.nb_filters):len(self.nb_filters) + 1]
        x2, blk2 = self._block_forward(a, B, T, 1, C, C, (2, 1), names, dropf, 300, nbt, train)
        x4, blk4 = self._block_forward(x2, B, T2, 1, C, C, (2, 1), names, dropf, 301, nbt, train)
        levels = []
        for seq, Tl, pfx, stream in ((a, T, "rnn", 200), (x2, T2, "rnn_2", 201), (x4, T4, "rnn_4", 202)):
            out, layers = self._gru_forward(seq.view(B, Tl, C), B, Tl, pfx, save)
            if drop > 0:
                out = ops.dropout(out, drop, stream, self.seed)
            levels.append((out, layers))
        g1, g2, g4 = (lv[0] for lv in levels)
        dev = x.device
        cat2 = torch.empty((B, T2, 512), device=dev, dtype=torch.float32)
        cat2[:, :, :256].copy_(g2)
        ops.upsample_time(g4, T2, out=cat2, out_offset=256)
        f2 = self._fuse(cat2, "conv1x1_2", B, T2)
        cat1 = torch.empty((B, T, 512), device=dev, dtype=torch.float32)
        cat1[:, :, :256].copy_(g1)
        ops.upsample_time(f2, T, out=cat1, out_offset=256)
        enc = self._fuse(cat1, "conv1x1_4", B, T)
        if save:
            ctx.update(T=T, T2=T2, T4=T4, drop=drop, fpn_blocks=(blk2, blk4), gru=[lv[1] for lv in levels],
                       cat1=cat1, cat2=cat2)
        return enc, ctx

    def run_backward(self, ctx, d_enc, on_early_grads=None):
        with ops.deferred_reductions():
            return self._run_backward_fpn(ctx, d_enc, on_early_grads)

    def _run_backward_fpn(self, ctx, d_enc, on_early_grads):
        B, T, T2, T4 = ctx["B"], ctx["T"], ctx["T2"], ctx["T4"]
        seed, drop = ctx["seed"], ctx["drop"]
        C = self.nb_filters[-1]
        d_cat1 = self._fuse_backward(ctx["cat1"], d_enc.contiguous(), "conv1x1_4", B, T)
        d_g1 = d_cat1[:, :, :256].contiguous()
        d_f2 = ops.upsample_time_bwd(d_cat1, T2, 256, in_offset=256)
        d_cat2 = self._fuse_backward(ctx["cat2"], d_f2, "conv1x1_2", B, T2)
        d_g2 = d_cat2[:, :, :256].contiguous()
        d_g4 = ops.upsample_time_bwd(d_cat2, T4, 256, in_offset=256)
        d_seq = []
        for d, Tl, pfx, stream, layers in ((d_g1, T, "rnn", 200, ctx["gru"][0]), (d_g2, T2, "rnn_2", 201, ctx["gru"][1]),
                                           (d_g4, T4, "rnn_4", 202, ctx["gru"][2])):
            if drop > 0:
                d = ops.dropout(d, drop, stream, seed)
            d_seq.append(self._gru_backward(layers, d, B, Tl, pfx))
        if not self.train_cnn:
            self._join_side()
            ops.flush_reductions()
            if on_early_grads is not None:
                on_early_grads()
            return
        blk2, blk4 = ctx["fpn_blocks"]
        d_x2 = self._block_backward(blk4, d_seq[2].view(B, T4, 1, C), B, seed)          # dL/d x_2 through level 4
        ops.axpy(d_x2.view(-1), d_seq[1].reshape(-1))                                    # + through rnn_2
        d_a = self._block_backward(blk2, d_x2.view(B, T2, 1, C), B, seed)                # dL/d a through level 2
        ops.axpy(d_a.view(-1), d_seq[0].reshape(-1))                                     # + through rnn
        self._cnn_backward(ctx, d_a.view(B, T, 1, C), on_early_grads)


class CRNN_pred(CRNN):
    """Drop-in for the reference's CNN-only tagger ``CRNN_pred`` (src/models/CRNN_GRL.py:206-290; BASELINE configs[1]):
    the CNN stack, then -- with the GRU commented out in the reference's forward -- ``strong = sigmoid(features)``,
    ``sof = clamp(softmax_class(dense_softmax(features)), 1e-7, 1)``, ``weak = sum_t strong*sof / sum_t sof``.
    ``forward(x, inference=False) -> (strong (B,T',C), weak (B,C))``.  The reference's shapes only agree for
    ``nclass == nb_filters[-1] == 2 * n_RNN_cell`` (the sigmoid acts on the 128 feature channels themselves), which is
    what is built.  ``rnn.*`` exists because the reference module owns those state-dict entries; no kernel reads it.
    Forward only (the reference never trains this module: no script instantiates it)."""
    _RUNS_RNN = False

    def __init__(self, n_in_channel, nclass, attention=False, activation="Relu", dropout=0, train_cnn=True,
                 rnn_type="BGRU", n_RNN_cell=64, n_layers_RNN=1, dropout_recurrent=0, cnn_integration=False,
                 learned_post=False, **cnn_kwargs):
        nb = list(cnn_kwargs.get("nb_filters", (64, 64, 64)))
        _check_cfg(nclass == nb[-1] == 2 * n_RNN_cell == 128, "CRNN_pred: nclass == nb_filters[-1] == 2*n_RNN_cell == 128")
        super().__init__(n_in_channel, nclass, attention, activation, dropout, train_cnn, rnn_type, n_RNN_cell,
                         n_layers_RNN, dropout_recurrent, cnn_integration, learned_post, **cnn_kwargs)

    def _extra_specs(self):
        C = self.nclass
        return [("dense_softmax.weight", (C, C)), ("dense_softmax.bias", (C,))], [], []

    def _init_roles(self):
        # reference module order (CRNN_GRL.py:211-236): dense_softmax is created before the CNN
        return [("dense_softmax", "linear")] + super()._init_roles()

    def run_forward(self, x, save=False):
        if x.dim() != 4 or x.shape[1] != 1:
            raise L.BsedError(f"CRNN_pred expects (B,1,T,F), got {tuple(x.shape)}")
        x = x.contiguous().float()
        B = x.shape[0]
        with ops.pack_cache(None if self.training else self._eval_plan):
            self._check_bf16_mode()
            a, T = self._cnn_forward(x, None)
            C = self.nclass
            feats = (a if a.dtype == torch.float32 else a.float()).view(B, T, C)   # bf16 mode: the head stays fp32
            w, b = self.P("dense_softmax.weight"), self.P("dense_softmax.bias")
            if self._mfma3:
                w3 = ops.pack_weight3(w, 1, C, C, 0, 1, C)
                logits, _ = ops.igemm3(feats, w3, C, 1, B * T, 1, C, ((0, 0),), bias=b)
            else:
                wpk = ops.pack_weight(w, 1, C, C, 0, 1, C)
                logits, _ = ops.igemm(feats, wpk, C, 1, B * T, 1, C, bias=b)
        return ops.tag_head_fwd(feats, logits.view(B, T, C))

    def run_backward(self, ctx, d):
        raise NotImplementedError("CRNN_pred is forward only (BASELINE configs[1]: CNN-only tagging forward)")

    def forward(self, x, inference=False):
        strong, weak = self.run_forward(x)
        if inference:
            # reference CRNN_GRL.py:282-287 (hard-codes 313 frames and .cuda() there; any T here)
            strong = strong * (weak > 0.5).float().unsqueeze(1)
        return strong, weak


class _CRNNFunction(torch.autograd.Function):
    """autograd bridge so reference-style drivers (loss.backward(); optimizer.step()) keep working."""

    @staticmethod
    def forward(ctx, x, module, flat):
        enc, c = module.run_forward(x, save=True)
        ctx.module, ctx.c = module, c
        return enc

    @staticmethod
    def backward(ctx, d_enc):
        ctx.module._attach_grads()
        ctx.module.run_backward(ctx.c, d_enc)
        ctx.c = None
        return None, None, None


class Predictor(_FlatModule):
    def __init__(self, nclass, attention=False, n_RNN_cell=64, device="cuda", **kwargs):
        super().__init__()
        L._require_gpu()
        _check_cfg(nclass == 20 and n_RNN_cell == 128, "nclass=20, n_RNN_cell=128")
        self.attention, self.nclass, self.K = attention, nclass, 2 * n_RNN_cell
        # dense / dense_softmax rows are adjacent: one (2C, K) matrix for the fused head kernel
        pspecs = [("dense.weight", (nclass, self.K)), ("dense_softmax.weight", (nclass, self.K)),
                  ("dense.bias", (nclass,)), ("dense_softmax.bias", (nclass,))]
        self._build(pspecs, [], device)
        self._init_order = [("dense", "linear")] + ([("dense_softmax", "linear")] if attention else [])
        if not attention:
            # the reference has no dense_softmax without attention: keep the tensors out of the state dict
            del self.dense_softmax
        self.reset_parameters()

    @torch.no_grad()
    def reset_parameters(self):
        b = 1 / math.sqrt(self.K)
        self.flat.uniform_(-b, b)

    def _wb(self):
        C, K = self.nclass, self.K
        return self.flat[:2 * C * K], self.flat[2 * C * K:]

    def run_forward(self, x):
        x = x.contiguous()
        B, T, K = x.shape
        w, b = self._wb()
        return ops.head_fwd(x, w, b, B, T, K, self.nclass, self.attention)

    def run_backward(self, x, saved, **loss_kw):
        """saved = (strong, sof, weak, den) from run_forward.  Accumulates dW/db, returns (dx, loss_part)."""
        B, T, K = x.shape
        C = self.nclass
        w, _ = self._wb()
        strong, sof, weak, den = saved
        dx, dw_part, db_part, loss_part = ops.head_bwd(x.contiguous(), w, strong.contiguous(), sof.contiguous(),
                                                       weak.contiguous(), den.contiguous(), B, T, K, C,
                                                       self.attention, **loss_kw)
        rows = dw_part.shape[0]                                   # B x time splits (bsed_head_splits)
        ops.reduce_partials(dw_part, rows, 1, 2 * C, K, 2 * C, K, self.flat_grad, 0, K, 1)
        ops.colsum(db_part, rows, 2 * C, 2 * C, self.flat_grad[2 * C * K:])
        return dx, loss_part

    def forward(self, x, inference=False):
        if torch.is_grad_enabled() and (x.requires_grad or self.training):
            strong, weak = _PredictorFunction.apply(x, self, self.dense.weight)
        else:
            strong, _, weak, _ = self.run_forward(x)
        if inference:
            # reference CRNN_GRL.py:452-457 (hard-codes 313 frames there; any T here)
            strong = strong * (weak > 0.5).float().unsqueeze(1)
        return strong, weak


class _PredictorFunction(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, module, flat):
        strong, sof, weak, den = module.run_forward(x)
        ctx.module = module
        ctx.save_for_backward(x, strong, sof, weak, den)
        return strong, weak

    @staticmethod
    def backward(ctx, g_strong, g_weak):
        x, strong, sof, weak, den = ctx.saved_tensors
        B, T, C = strong.shape
        gs = g_strong.contiguous() if g_strong is not None else torch.zeros_like(strong)
        gw = g_weak.contiguous() if g_weak is not None else torch.zeros_like(weak)
        ctx.module._attach_grads()
        dx, _ = ctx.module.run_backward(x, (strong, sof, weak, den), g_strong=gs, g_weak=gw, w_strong=0.0,
                                        w_weak=0.0)
        return dx, None, None


def weights_init(m):
    """Reference ``weights_init`` (src/utilities/utils.py:40-63) for a bsed_amd module: xavier-uniform(gain sqrt 2)
    convolutions with zero bias, BatchNorm ~ N(1, 0.02) / 0, orthogonal GRU matrices (biases untouched), N(0, 0.01)
    Linear weights with zero bias (this includes the GLU's Linear).  The role of every tensor comes from the module's
    ``_init_order`` table (what the tensor IS), and the draws are made on the CPU generator in the order
    ``reference_model.apply(weights_init)`` makes them, so the same ``torch.manual_seed`` gives the reference's weights."""
    order = getattr(m, "_init_order", None)
    if order is None:
        return
    with torch.no_grad():
        for name, role in order:
            if role == "gru":
                # the reference matches class names by substring: apply() reaches nn.GRU ("GRU") and then its wrapper
                # BidirectionalGRU (also contains "GRU"), so every matrix is drawn twice and the second draw stays
                for _ in range(2):
                    for mat in m._gru_matrices(name):
                        p = m.P(mat)
                        t = torch.empty(p.shape)
                        nn.init.orthogonal_(t)
                        p.copy_(t)
                continue
            w, b = m.P(name + ".weight"), m.P(name + ".bias")
            if role == "conv":
                t = torch.empty(w.shape)
                nn.init.xavier_uniform_(t, gain=np.sqrt(2))
            elif role == "bn":
                t = torch.empty(w.shape).normal_(1.0, 0.02)
            elif role == "linear":
                t = torch.empty(w.shape).normal_(0, 0.01)
            else:
                raise L.BsedError(f"weights_init: unknown role {role!r} for {name}")
            w.copy_(t)
            b.zero_()
