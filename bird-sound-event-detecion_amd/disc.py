"""Domain-adversarial head on the GPU.

  Clip_Discriminator(input_dim, dropout=0)            <- reference src/models/CRNN_GRL.py:16-53
  ConditionalDomainAdversarialLoss(discriminator)      <- reference src/DA/cdan_frame.py:16-119 as it executes:
        d = D(GRL_lambda(cat(f_s, f_t)));  loss = BCE(d, [1]*B_s + [0]*B_t)
        lambda_i = 2/(1+exp(-i/1000)) - 1, i += 1 per call   (WarmStartGradientReverseLayer, src/DA/grl.py:33-73)
Layers 2 and 3 (98 % of the multiply-adds) run directly on the implicit-GEMM kernels through a space-to-depth
re-layout (a 3x3 / stride-2 convolution is a 2x2 / stride-1 one over 4C channels); layers 1, 4 and 5 (1, 32, 16 input
channels) keep the im2col lowering (csrc/disc.hip, csrc/igemm.hip, csrc/igemm3.hip).
"""
import ctypes
import math

import numpy as np
import torch
import torch.nn as nn

from . import _lib as L
from . import ops
from .models import _FlatModule

D_EPS, D_MOM = 1e-5, 0.1
D_CH = [1, 128, 64, 32, 16, 8]


def _i(v):
    return ctypes.c_int(v)


def grl_coeff(iter_num, alpha=1.0, lo=0.0, hi=1.0, max_iters=1000.0):
    """WarmStartGradientReverseLayer coefficient at call ``iter_num`` (reference src/DA/grl.py:62-67)"""
    return float(2.0 * (hi - lo) / (1.0 + np.exp(-alpha * iter_num / max_iters)) - (hi - lo) + lo)


def _im2col(act, scale, shift, N, Hi, Wi, C, CP):
    Ho, Wo = (Hi - 3) // 2 + 1, (Wi - 3) // 2 + 1
    K = 16 if C == 1 else 9 * CP
    col = torch.empty((N * Ho * Wo, K), device=act.device, dtype=torch.float32)
    L.call("bsed_im2col_s2", L.ptr(act), L.ptr(scale), L.ptr(shift), L.ptr(col), _i(N), _i(Hi), _i(Wi), _i(C), _i(CP),
           L.stream())
    return col, Ho, Wo, K


def _col2im(dcol, y, scale, shift, N, Hi, Wi, C, CP, out_scale=1.0):
    dev = dcol.device
    out = torch.empty((N, Hi, Wi, C), device=dev, dtype=torch.float32)
    stats = None
    if C > 1:
        nb = L.lib().bsed_col2im_s2_num_blocks(N, Hi, Wi, C)
        stats = torch.empty((nb, 2, C), device=dev, dtype=torch.float32)
    L.call("bsed_col2im_s2", L.ptr(dcol), L.ptr(y), L.ptr(scale), L.ptr(shift), L.ptr(out), L.ptr(stats), _i(N), _i(Hi),
           _i(Wi), _i(C), _i(CP), ctypes.c_float(out_scale), L.stream())
    return out, stats


TAPS2x2 = [(0, 0), (0, 1), (1, 0), (1, 1)]
# weight slot ((dp*2+dq)*2+a)*2+b of the space-to-depth form holds W[kh = 2dq+b][kw = 2dp+a] (image rows = time pair
# with the reference's kw, image columns = features with its kh: the image is kept in the embedding's (T, 256) order)
_S2D_SLOT = [(((kw >> 1) * 2 + (kh >> 1)) * 2 + (kw & 1)) * 2 + (kh & 1) for kh in range(3) for kw in range(3)]
DIRECT_LAYERS = (2, 3)


def _s2d_fwd(y, scale, shift, N, Ha, Wa, Hi, Wi, C):
    Hp, Wp = (Hi + 1) // 2, (Wi + 1) // 2
    xp = torch.empty((N, Hp, Wp, 4 * C), device=y.device, dtype=torch.float32)
    ops._note("s2d_fwd_kernel", f"{Hi}x{Wi}x{C}", 3.0 * xp.numel(), 4.0 * (N * Hi * Wi * C + xp.numel()))
    L.call("bsed_s2d_fwd", L.ptr(y), L.ptr(scale), L.ptr(shift), L.ptr(xp), _i(N), _i(Ha), _i(Wa), _i(Hi), _i(Wi), _i(C),
           L.stream())
    return xp, Hp, Wp


def _s2d_bwd(dxp, y, scale, shift, N, Ha, Wa, Hi, Wi, C):
    g = torch.empty((N, Ha, Wa, C), device=y.device, dtype=torch.float32)
    nb = L.lib().bsed_s2d_num_blocks(N, Ha, Wa, C)
    stats = torch.empty((nb, 2, C), device=y.device, dtype=torch.float32)
    ops._note("s2d_bwd_kernel", f"{Hi}x{Wi}x{C}", 4.0 * g.numel(), 4.0 * (N * Hi * Wi * C + 2 * g.numel()))
    L.call("bsed_s2d_bwd", L.ptr(dxp), L.ptr(y), L.ptr(scale), L.ptr(shift), L.ptr(g), L.ptr(stats), _i(N), _i(Ha), _i(Wa),
           _i(Hi), _i(Wi), _i(C), L.stream())
    return g, stats


class Clip_Discriminator(_FlatModule):
    def __init__(self, input_dim=None, dropout=0, device="cuda"):
        super().__init__()
        L._require_gpu()
        pspecs, bspecs = [], []
        for k in range(1, 6):
            pspecs += [(f"conv_{k}.weight", (D_CH[k], D_CH[k - 1], 3, 3)), (f"conv_{k}.bias", (D_CH[k],))]
        pspecs += [("dense_d.weight", (1, 16)), ("dense_d.bias", (1,))]
        for k in range(1, 6):
            pspecs += [(f"bn_{k}.weight", (D_CH[k],)), (f"bn_{k}.bias", (D_CH[k],))]
            bspecs += [(f"bn_{k}.running_mean", (D_CH[k],)), (f"bn_{k}.running_var", (D_CH[k],))]
        self._build(pspecs, bspecs, device)
        import os
        # GEMM family of the convolutions: "fp32" (fp32 matrix cores, the default) or "bf16x3" (split-fp32 operands on
        # the bf16 cores, ~4 % faster adversarial step).  Unlike the CRNN (smooth GLU gates) this network has LeakyReLU
        # between its layers: a forward rounding error eps flips the sign of ~eps of the pre-activations, each flip
        # changes one backward mask element by 0.8, so gradient errors go like sqrt(eps) -- measured against an fp64
        # oracle at 24 x 216 x 256: 6e-4 with fp32 GEMMs (what any fp32 implementation gets), 8e-3 with split-fp32.
        # Loss and outputs agree to 3e-7 either way; parity of the gradients is why fp32 is the default here.
        self.conv_mode = os.environ.get("BSED_DISC_MODE", "fp32")
        # The mask sensitivity is a property of the FORWARD contractions only (the masks are functions of the forward
        # pre-activations): the data-gradient GEMMs of the backward pass contract given operands, where split-fp32 costs
        # its usual 2^-16 and nothing is amplified.  They therefore run on the bf16 cores by default (the weight
        # gradients always did): -2 ms per adversarial step at B = 128 + 128.  BSED_DISC_BWD_MODE=fp32 for A/B runs.
        self.bwd_mode = os.environ.get("BSED_DISC_BWD_MODE", "bf16x3")
        self.nbt = torch.zeros(5, device=device, dtype=torch.int64)
        for k in range(1, 6):
            self.P(f"bn_{k}").register_buffer("num_batches_tracked", self.nbt[k - 1])
        self.reset_parameters()

    @torch.no_grad()
    def reset_parameters(self):
        for name, p in self.named_parameters():
            if name.startswith("bn_"):
                p.fill_(1.0 if name.endswith("weight") else 0.0)
            elif name.endswith("weight"):
                b = 1 / math.sqrt(p[0].numel())
                p.uniform_(-b, b)
            else:
                b = 1 / math.sqrt(self.P(name[:-4] + "weight")[0].numel())
                p.uniform_(-b, b)
        for name, b in self.named_buffers():
            if name.endswith("running_var"):
                b.fill_(1.0)
            else:
                b.zero_()

    # ------------------------------------------------------------------ forward / backward
    def _fwd_weight(self, k):
        """[K][NP] GEMM operand of conv_k: row (dw*3+dh)*CP + c  (== kh*3+kw of the reference orientation)"""
        co, cin = D_CH[k], D_CH[k - 1]
        w = self.P(f"conv_{k}.weight")
        wpk = ops.pack_weight(w, 9, cin, co, 1, 9, cin * 9)            # [9][cin][NP]
        NP = wpk.shape[2]
        if cin == 1:
            full = torch.zeros((1, 16, NP), device=w.device, dtype=torch.float32)
            full[0, :9] = wpk[:, 0]
            return full, 1
        CP = max(cin, 32)
        if CP != cin:
            full = torch.zeros((9, CP, NP), device=w.device, dtype=torch.float32)
            full[:, :cin] = wpk
            wpk = full
        return wpk.view(1, 9 * CP, NP), CP

    def _bwd_weight(self, k, CP):
        """[co_pad][NP] operand of dcol = dY @ W^T: column (dw*3+dh)*CP + c"""
        co, cin = D_CH[k], D_CH[k - 1]
        w = self.P(f"conv_{k}.weight").detach()
        K = 16 if cin == 1 else 9 * CP
        cop = ops.round_up(co, 16)
        full = torch.zeros((1, cop, ops.round_up(K, 32)), device=w.device, dtype=torch.float32)
        if cin == 1:
            full[0, :co, :9] = w.view(co, 9)
        else:
            wt = w.view(co, cin, 9).permute(0, 2, 1)                    # (co, 9, cin)
            tmp = torch.zeros((co, 9, CP), device=w.device, dtype=torch.float32)
            tmp[:, :, :cin] = wt
            full[0, :co, :K] = tmp.view(co, K)
        return full, K, cop

    def _s2d_weight(self, k):
        """(4 taps, 4*cin, co) weight of the space-to-depth form of conv_k: slot (dp,dq,a,b) = W[2dq+b][2dp+a] or zero"""
        co, cin = D_CH[k], D_CH[k - 1]
        w = self.P(f"conv_{k}.weight").detach()
        full = torch.zeros((16, cin, co), device=w.device, dtype=torch.float32)
        full[_S2D_SLOT] = w.permute(2, 3, 1, 0).reshape(9, cin, co)
        return full.view(4, 4 * cin, co)

    def run_forward(self, feat, n_source=None, save=True):
        """feat (N,T,256) -> (d (N,), ctx).  In train mode (and with n_source) also prepares the BCE backward."""
        feat = feat.contiguous().float()
        N, T, F = feat.shape
        if F != 256:
            raise L.BsedError(f"Clip_Discriminator is built for 256-feature encodings (2 x 128 GRU cells), got {F}")
        train = self.training
        # act: pre-BatchNorm output of the previous layer on an ALLOCATED grid (Ha, Wa) whose VALID extent is (Hi, Wi)
        act, scale, shift = feat, None, None
        Ha, Wa, Hi, Wi = T, F, T, F
        layers = []
        for k in range(1, 6):
            co, cin = D_CH[k], D_CH[k - 1]
            epi = ops.EPI_STATS if train else ops.EPI_PLAIN
            bias = self.P(f"conv_{k}.bias")
            Ho, Wo = (Hi - 3) // 2 + 1, (Wi - 3) // 2 + 1
            M = N * Ho * Wo
            if k in DIRECT_LAYERS:
                xp, Hp, Wp = _s2d_fwd(act, scale, shift, N, Ha, Wa, Hi, Wi, cin)
                assert (Hp - 1, Wp - 1) == (Ho, Wo)
                wfull = self._s2d_weight(k)
                K = 4 * cin
                if self.conv_mode == "bf16x3":
                    w3 = ops.pack_weight3(wfull, 4, K, co, K * co, co, 1)
                    y, stats = ops.igemm3(xp, w3, co, N, Hp, Wp, K, TAPS2x2, bias=bias, epilogue=epi, valid=(Ho, Wo))
                else:
                    wpk = ops.pack_weight(wfull, 4, K, co, K * co, co, 1)
                    y, stats = ops.igemm(xp, wpk, co, N, Hp, Wp, K, taps=TAPS2x2, bias=bias, epilogue=epi, valid=(Ho, Wo))
                rec = dict(direct=True, xp=xp, wfull=wfull, Hp=Hp, Wp=Wp)
                nHa, nWa = Hp, Wp
            else:
                if (Ha, Wa) != (Hi, Wi):  # compact copy of the valid extent for the im2col gather (small layers only)
                    act = act[:, :Hi, :Wi, :].contiguous()
                wpk, CP = self._fwd_weight(k)
                col, Ho_, Wo_, K = _im2col(act, scale, shift, N, Hi, Wi, cin, CP)
                if self.conv_mode == "bf16x3" and K % 32 == 0:
                    w3 = ops.pack_weight3(wpk, 1, K, co, 0, wpk.shape[2], 1)
                    y, stats = ops.igemm3(col, w3, co, 1, M, 1, K, ((0, 0),), bias=bias, epilogue=epi)
                else:
                    y, stats = ops.igemm(col, wpk, co, 1, M, 1, K, bias=bias, epilogue=epi)
                y = y.view(N, Ho, Wo, co)
                rec = dict(direct=False, col=col, K=K, CP=CP)
                nHa, nWa = Ho, Wo
            bn = self.P(f"bn_{k}")
            if train:
                mean, invstd, nscale, nshift = ops.bn_finalize(stats, co, float(M), D_EPS, D_MOM, bn.weight, bn.bias,
                                                               bn.running_mean, bn.running_var, self.nbt[k - 1:k])
            else:
                mean = invstd = None
                nscale, nshift = ops.bn_eval(co, D_EPS, bn.weight, bn.bias, bn.running_mean, bn.running_var)
            if save:
                rec.update(y=y, mean=mean, invstd=invstd, scale=nscale, shift=nshift, Ha_in=Ha, Wa_in=Wa, Hi=Hi, Wi=Wi,
                           Ho=Ho, Wo=Wo, Ha=nHa, Wa=nWa, M=M)
                layers.append(rec)
            act, scale, shift = y, nscale, nshift
            Ha, Wa, Hi, Wi = nHa, nWa, Ho, Wo
        dev = feat.device
        d = torch.empty((N,), device=dev, dtype=torch.float32)
        do_loss = train and n_source is not None
        g5 = stats5 = dwl = dbl = lossp = None
        if do_loss:
            g5 = torch.empty_like(act)
            stats5 = torch.empty((N, 2, 8), device=dev, dtype=torch.float32)
            dwl = torch.empty((N, 2, 16), device=dev, dtype=torch.float32)
            dbl = torch.empty((N, 2, 1), device=dev, dtype=torch.float32)
            lossp = torch.empty((N, 2, 1), device=dev, dtype=torch.float32)
        dense = self.P("dense_d")
        L.call("bsed_disc_head", L.ptr(act), L.ptr(scale), L.ptr(shift), ctypes.c_void_p(dense.weight.data_ptr()),
               ctypes.c_void_p(dense.bias.data_ptr()), _i(N), _i(n_source or 0), _i(Hi), _i(Wi), _i(8),
               _i(1 if do_loss else 0), L.ptr(d), L.ptr(g5), L.ptr(stats5), L.ptr(dwl), L.ptr(dbl), L.ptr(lossp),
               L.stream())
        ctx = dict(N=N, T=T, F=F, layers=layers, g5=g5, stats5=stats5, dwl=dwl, dbl=dbl, lossp=lossp) if save else None
        return d, ctx

    def run_backward(self, ctx, grl_coeff):
        """Accumulates the discriminator's parameter gradients; returns dL/d feat (N,T,256) already multiplied by
        -grl_coeff (the gradient-reverse layer)."""
        with ops.deferred_reductions():   # the weight / bias gradient reductions go out in one launch at the end
            return self._run_backward(ctx, grl_coeff)

    def _run_backward(self, ctx, grl_coeff):
        N = ctx["N"]
        lay = ctx["layers"]
        dense = self.P("dense_d")
        ops.stats_to_grad(ctx["dwl"], 16, 0, dense.weight.grad)
        ops.stats_to_grad(ctx["dbl"], 1, 0, dense.bias.grad)
        # g: dL/d(BatchNorm_k output pre-activation side), on layer k's allocated grid; stats: its (sum g, sum g*y) partials
        g, stats = ctx["g5"], ctx["stats5"]
        for k in range(5, 0, -1):
            l = lay[k - 1]
            co, cin = D_CH[k], D_CH[k - 1]
            bn = self.P(f"bn_{k}")
            ops.bn_bwd(stats, co, float(l["M"]), bn.weight, l["mean"], l["invstd"], bn.weight.grad, bn.bias.grad, g,
                       l["y"])
            w = self.P(f"conv_{k}.weight")
            p = lay[k - 2] if k > 1 else None
            if l["direct"]:
                Hp, Wp, K = l["Hp"], l["Wp"], 4 * cin
                dy = g.view(N, Hp, Wp, co)
                # the BatchNorm-backward map is affine: it left non-zero values on the grid's non-output row / column
                dy[:, l["Ho"]:, :, :] = 0
                dy[:, :, l["Wo"]:, :] = 0
                part, G, KP, NP = ops.wgrad(l["xp"], dy, N, Hp, Wp, K, co, taps=TAPS2x2)
                tmp = torch.empty((4, K, co), device=dy.device, dtype=torch.float32)
                ops.reduce_partials(part, G, 4, KP, NP, K, co, tmp, K * co, co, 1, accumulate=False, defer=False)
                w.grad.add_(tmp.view(16, cin, co)[_S2D_SLOT].permute(2, 1, 0).reshape(co, cin, 3, 3))
                flipped = [(-a, -b) for a, b in TAPS2x2]
                if self.conv_mode == "bf16x3" or self.bwd_mode == "bf16x3":
                    wd3 = ops.pack_weight3(l["wfull"], 4, co, K, K * co, 1, co)
                    dxp, _ = ops.igemm3(dy, wd3, K, N, Hp, Wp, co, flipped)
                else:
                    wd = ops.pack_weight(l["wfull"], 4, co, K, K * co, 1, co)
                    dxp, _ = ops.igemm(dy, wd, K, N, Hp, Wp, co, taps=flipped)
                g, stats = _s2d_bwd(dxp, p["y"], p["scale"], p["shift"], N, l["Ha_in"], l["Wa_in"], l["Hi"], l["Wi"], cin)
                continue
            dy = g.view(l["M"], co)
            part, G, KP, NP = ops.wgrad(l["col"], dy, 1, l["M"], 1, l["K"], co)
            if cin == 1:
                ops.reduce_partials(part, G, 1, KP, NP, 9, co, w.grad, 0, 1, 9)
            else:
                assert KP == 9 * l["CP"]
                ops.reduce_partials(part, G, 9, l["CP"], NP, cin, co, w.grad, 1, 9, cin * 9)
            wT, K, cop = self._bwd_weight(k, l["CP"])
            if cop != co:  # the GEMM contracts over multiples of 16 channels: zero-pad dY of the last layer
                dyp = torch.zeros((l["M"], cop), device=dy.device, dtype=torch.float32)
                dyp[:, :co] = dy
                dy = dyp
            if (self.conv_mode == "bf16x3" or self.bwd_mode == "bf16x3") and cop % 32 == 0:
                w3 = ops.pack_weight3(wT, 1, cop, K, 0, wT.shape[2], 1)
                dcol, _ = ops.igemm3(dy, w3, K, 1, l["M"], 1, cop, ((0, 0),))
            else:
                dcol, _ = ops.igemm(dy, wT, K, 1, l["M"], 1, cop)
            if k > 1:
                # gradient on the compact (Hi, Wi) extent of the previous layer's output ...
                yc = p["y"] if (p["Ha"], p["Wa"]) == (l["Hi"], l["Wi"]) else p["y"][:, :l["Hi"], :l["Wi"], :].contiguous()
                gc, stats = _col2im(dcol, yc, p["scale"], p["shift"], N, l["Hi"], l["Wi"], cin, l["CP"])
                if (p["Ha"], p["Wa"]) == (l["Hi"], l["Wi"]):
                    g = gc
                else:  # ... put back on that layer's allocated grid (zero on its non-output row / column)
                    g = torch.zeros((N, p["Ha"], p["Wa"], cin), device=gc.device, dtype=torch.float32)
                    g[:, :l["Hi"], :l["Wi"], :] = gc
            else:
                dfeat, _ = _col2im(dcol, None, None, None, N, l["Hi"], l["Wi"], 1, 1, out_scale=-float(grl_coeff))
        return dfeat.view(N, ctx["T"], ctx["F"])

    def forward(self, x):
        d, _ = self.run_forward(x, save=False)
        return d.view(-1, 1)


class Frame_Discriminator(_FlatModule):
    """Drop-in for the reference's frame-level discriminator (src/models/CRNN_GRL.py:116-140):
    ``d = sigmoid(L3(drop(leaky(L2(drop(leaky(L1(x))))))))`` per frame, (N,T,256) -> (N,T,1), LeakyReLU(0.2), the two wide
    layers as 1-tap contractions on the implicit-GEMM kernels, the 32 -> 1 head and the elementwise stages in
    csrc/disc.hip.  ``disc(x)`` is differentiable (autograd bridge); ``run_forward`` / ``run_backward`` are the explicit
    pair.  The reference never gives this module a runnable loss (main_baseline.py:789-796 pairs it with a loss whose
    labels are per clip: torch's BCE rejects the shapes) -- parity is the module's own forward and
    backward."""
    SLOPE = 0.2

    def __init__(self, input_dim=None, dropout=0, device="cuda"):
        super().__init__()
        L._require_gpu()
        self.dropout_p = float(dropout)
        self.seed = 0
        import os
        self.conv_mode = os.environ.get("BSED_DISC_MODE", "fp32")
        pspecs = [("dense_d_1.weight", (128, 256)), ("dense_d_1.bias", (128,)),
                  ("dense_d_2.weight", (32, 128)), ("dense_d_2.bias", (32,)),
                  ("dense_d_3.weight", (1, 32)), ("dense_d_3.bias", (1,))]
        self._build(pspecs, [], device)
        self._init_order = [("dense_d_1", "linear"), ("dense_d_2", "linear"), ("dense_d_3", "linear")]
        self.reset_parameters()

    @torch.no_grad()
    def reset_parameters(self):
        for name, p in self.named_parameters():
            w = p if name.endswith("weight") else self.P(name[:-4] + "weight")
            b = 1 / math.sqrt(w.shape[1])
            p.uniform_(-b, b)

    def set_seed(self, seed):
        self.seed = int(seed)

    def _linear(self, x, name, M, K, N):
        w, b = self.P(name + ".weight"), self.P(name + ".bias")
        if self.conv_mode == "bf16x3":
            return ops.igemm3(x, ops.pack_weight3(w, 1, K, N, 0, 1, K), N, 1, M, 1, K, ((0, 0),), bias=b)[0]
        return ops.igemm(x, ops.pack_weight(w, 1, K, N, 0, 1, K), N, 1, M, 1, K, bias=b)[0]

    def _linear_bwd(self, x, dy, name, M, K, N):
        """accumulates dW, db of y = x W^T + b; returns dL/dx (M,K)"""
        w, b = self.P(name + ".weight"), self.P(name + ".bias")
        ops.colsum(dy, M, N, N, b.grad)
        part, G, KP, NP = ops.wgrad(x, dy, 1, M, 1, K, N)
        ops.reduce_partials(part, G, 1, KP, NP, K, N, w.grad, 0, 1, K)
        if self.conv_mode == "bf16x3":
            return ops.igemm3(dy, ops.pack_weight3(w, 1, N, K, 0, K, 1), K, 1, M, 1, N, ((0, 0),))[0]
        return ops.igemm(dy, ops.pack_weight(w, 1, N, K, 0, K, 1), K, 1, M, 1, N)[0]

    def _act(self, a, stream_id, drop):
        out = torch.empty_like(a)
        L.call("bsed_leaky_dropout_fwd", L.ptr(a), L.ptr(out), ctypes.c_long(a.numel()), ctypes.c_float(self.SLOPE),
               ctypes.c_float(drop), ctypes.c_uint32(stream_id), ctypes.c_uint64(self.seed), L.stream())
        return out

    def _act_bwd(self, d_out, a, stream_id, drop, seed):
        d_a = torch.empty_like(a)
        L.call("bsed_leaky_dropout_bwd", L.ptr(d_out), L.ptr(a), L.ptr(d_a), ctypes.c_long(a.numel()),
               ctypes.c_float(self.SLOPE), ctypes.c_float(drop), ctypes.c_uint32(stream_id), ctypes.c_uint64(seed), L.stream())
        return d_a

    def run_forward(self, x, save=True):
        x = x.contiguous().float()
        N, T, F = x.shape
        if F != 256:
            raise L.BsedError(f"Frame_Discriminator takes 256-feature encodings, got {F}")
        M = N * T
        drop = self.dropout_p if self.training else 0.0
        x2d = x.view(M, 256)
        a1 = self._linear(x2d, "dense_d_1", M, 256, 128).view(M, 128)
        h1 = self._act(a1, 401, drop)
        a2 = self._linear(h1, "dense_d_2", M, 128, 32).view(M, 32)
        h2 = self._act(a2, 402, drop)
        d = torch.empty((M,), device=x.device, dtype=torch.float32)
        w3, b3 = self.P("dense_d_3.weight"), self.P("dense_d_3.bias")
        L.call("bsed_frame_head_fwd", L.ptr(h2), ctypes.c_void_p(w3.data_ptr()), ctypes.c_void_p(b3.data_ptr()), L.ptr(d),
               ctypes.c_long(M), _i(32), L.stream())
        ctx = dict(x=x2d, a1=a1, h1=h1, a2=a2, h2=h2, d=d, M=M, drop=drop, seed=self.seed, shape=(N, T)) if save else None
        return d.view(N, T, 1), ctx

    def run_backward(self, ctx, d_out):
        """d_out: dL/d(domain_out) (N,T,1).  Accumulates parameter gradients, returns dL/dx (N,T,256)."""
        M = ctx["M"]
        w3, b3 = self.P("dense_d_3.weight"), self.P("dense_d_3.bias")
        G = int(min(1024, max(1, M // 256)))
        dh2 = torch.empty((M, 32), device=d_out.device, dtype=torch.float32)
        part = torch.empty((G, 2, 32), device=d_out.device, dtype=torch.float32)
        L.call("bsed_frame_head_bwd", L.ptr(ctx["h2"]), ctypes.c_void_p(w3.data_ptr()), L.ptr(ctx["d"]),
               L.ptr(d_out.contiguous().view(M)), L.ptr(dh2), L.ptr(part), _i(G), ctypes.c_long(M), _i(32), L.stream())
        ops.stats_to_grad(part, 32, 0, w3.grad)
        tmp = torch.zeros(32, device=d_out.device, dtype=torch.float32)
        ops.stats_to_grad(part, 32, 1, tmp)
        ops.axpy(b3.grad, tmp[:1])
        da2 = self._act_bwd(dh2, ctx["a2"], 402, ctx["drop"], ctx["seed"])
        dh1 = self._linear_bwd(ctx["h1"], da2, "dense_d_2", M, 128, 32).view(M, 128)
        da1 = self._act_bwd(dh1, ctx["a1"], 401, ctx["drop"], ctx["seed"])
        dx = self._linear_bwd(ctx["x"], da1, "dense_d_1", M, 256, 128)
        N, T = ctx["shape"]
        return dx.view(N, T, 256)

    def forward(self, x):
        if torch.is_grad_enabled() and (x.requires_grad or self.training):
            return _FrameDFunction.apply(x, self, self.P("dense_d_1.weight"))
        return self.run_forward(x, save=False)[0]


class _FrameDFunction(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, module, flat):
        d, c = module.run_forward(x, save=True)
        ctx.module, ctx.c = module, c
        return d

    @staticmethod
    def backward(ctx, d_out):
        ctx.module._attach_grads()
        dx = ctx.module.run_backward(ctx.c, d_out)
        ctx.c = None
        return dx, None, None


class ConditionalDomainAdversarialLoss(nn.Module):
    """Same call as the reference: ``loss = cdan(g_s, f_s, g_t, f_t)`` (g_* are accepted and, as in the reference's
    active code path, do not enter the result).  ``forward`` returns the loss as a device scalar and stores what
    ``backward_features()`` needs; the explicit pair is what SEDTrainer uses."""

    def __init__(self, domain_discriminator, entropy_conditioning=False, randomized=False, num_classes=-1,
                 features_dim=-1, randomized_dim=1024, reduction="mean"):
        super().__init__()
        if entropy_conditioning or randomized or reduction != "mean":
            raise NotImplementedError("only the reference's active configuration (plain BCE, mean) is built")
        if isinstance(domain_discriminator, Frame_Discriminator):
            # the reference's own pairing (main_baseline.py:789-796) dies in F.binary_cross_entropy: (N,T,1) predictions
            # against (N,) clip labels (src/DA/cdan_frame.py:99-103,119); same error behaviour here
            raise ValueError("Using a target size (torch.Size([N])) that is different to the input size "
                             "(torch.Size([N, T, 1])) is deprecated. Please ensure they have the same size. "
                             "[ConditionalDomainAdversarialLoss labels clips; Frame_Discriminator scores frames]")
        self.domain_discriminator = domain_discriminator
        self.iter_num = 0
        self.alpha, self.lo, self.hi, self.max_iters = 1.0, 0.0, 1.0, 1000.0
        self._ctx = None

    def grl_coeff(self):
        return grl_coeff(self.iter_num, self.alpha, self.lo, self.hi, self.max_iters)

    def forward(self, g_s, f_s, g_t, f_t):
        coeff = self.grl_coeff()
        self.iter_num += 1  # auto_step
        f = torch.cat((f_s.detach(), f_t.detach()), 0)
        d, ctx = self.domain_discriminator.run_forward(f, n_source=f_s.shape[0], save=True)
        ctx["coeff"], ctx["ns"] = coeff, f_s.shape[0]
        self._ctx = ctx
        self.domain_out = d
        return ctx["lossp"][:, 0, 0].sum() / f.shape[0]

    def backward_features(self):
        """(d_f_s, d_f_t) = gradient of the domain loss w.r.t. the features THROUGH the gradient-reverse layer;
        the discriminator's own parameter gradients are accumulated into its flat_grad."""
        ctx, self._ctx = self._ctx, None
        df = self.domain_discriminator.run_backward(ctx, ctx["coeff"])
        return df[:ctx["ns"]], df[ctx["ns"]:]
