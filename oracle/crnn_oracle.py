"""CRNN / Predictor / train-step oracle on stock ``torch.nn`` (CPU fp32).
TEST INFRASTRUCTURE ONLY -- never on the product path.

Restates the reference's module graph with identical ``state_dict`` key names so the
reference's own modules (importable in the build container only) and this oracle
can exchange weights; pinned by ``tests/golden/crnn_*.npz`` (oracle/gen_golden.py).

  * GLU / CNN            <- /root/reference/src/models/CNN.py:5-16, 33-84
  * BidirectionalGRU     <- src/models/RNN.py:7-16
  * CRNN                 <- src/models/CRNN_GRL.py:142-204
  * Predictor            <- src/models/CRNN_GRL.py:430-460
  * CRNN_pred            <- src/models/CRNN_GRL.py:206-290 (CNN-only tagger, BASELINE configs[1])
  * Clip_Discriminator   <- src/models/CRNN_GRL.py:16-53
  * Frame_Discriminator  <- src/models/CRNN_GRL.py:116-140
  * weights_init         <- src/utilities/utils.py:40-63
  * update_ema_variables <- src/main_baseline.py:91-105
  * ramps                <- src/utilities/ramps.py:4-30
  * train-step loss      <- src/main_baseline.py:431-529 (no ISP), adversarial term
                            src/main_scmt_ada_weak.py:312-339,527-528 with
                            src/DA/cdan_frame.py:89-119 + src/DA/grl.py:12-73
"""
import math
from collections import OrderedDict

import numpy as np
import torch
import torch.nn as nn
import torch.nn.functional as F

CRNN_KWARGS = dict(
    n_in_channel=1, nclass=20, attention=True, n_RNN_cell=128, n_layers_RNN=2,
    activation="glu", dropout=0.5, kernel_size=7 * [3], padding=7 * [1], stride=7 * [1],
    nb_filters=[16, 32, 64, 128, 128, 128, 128],
    pooling=[[2, 2], [2, 2], [1, 2], [1, 2], [1, 2], [1, 2], [1, 2]],
)
PREDICTOR_KWARGS = dict(nclass=20, attention=True, n_RNN_cell=128)


class GLU(nn.Module):
    def __init__(self, c):
        super().__init__()
        self.sigmoid = nn.Sigmoid()
        self.linear = nn.Linear(c, c)

    def forward(self, x):  # x: (B,C,T,F)
        lin = self.linear(x.permute(0, 2, 3, 1)).permute(0, 3, 1, 2)
        return lin * self.sigmoid(x)


class CNN(nn.Module):
    def __init__(self, n_in_channel, activation="Relu", conv_dropout=0, kernel_size=(3, 3, 3),
                 padding=(1, 1, 1), stride=(1, 1, 1), nb_filters=(64, 64, 64),
                 pooling=((1, 4), (1, 4), (1, 4))):
        super().__init__()
        self.nb_filters = list(nb_filters)
        layers = OrderedDict()
        for i, n_out in enumerate(nb_filters):
            n_in = n_in_channel if i == 0 else nb_filters[i - 1]
            layers[f"conv{i}"] = nn.Conv2d(n_in, n_out, kernel_size[i], stride[i], padding[i])
            layers[f"batchnorm{i}"] = nn.BatchNorm2d(n_out, eps=0.001, momentum=0.99)
            act = activation.lower()
            if act == "glu":
                layers[f"glu{i}"] = GLU(n_out)
            elif act == "relu":
                layers[f"relu{i}"] = nn.ReLU()
            elif act == "leakyrelu":
                layers[f"relu{i}"] = nn.LeakyReLU(0.2)
            else:
                raise NotImplementedError(activation)
            if conv_dropout is not None:
                layers[f"dropout{i}"] = nn.Dropout(conv_dropout)
            layers[f"pooling{i}"] = nn.AvgPool2d(tuple(pooling[i]))
        self.cnn = nn.Sequential(layers)

    # the reference strips one "cnn." level from the keys (CNN.py:71-75)
    def state_dict(self, *a, **k):
        return self.cnn.state_dict(*a, **k)

    def load_state_dict(self, sd, strict=True):
        return self.cnn.load_state_dict(sd, strict)

    def forward(self, x):
        return self.cnn(x)


class BidirectionalGRU(nn.Module):
    def __init__(self, n_in, n_hidden, dropout=0, num_layers=1):
        super().__init__()
        self.rnn = nn.GRU(n_in, n_hidden, bidirectional=True, dropout=dropout, batch_first=True,
                          num_layers=num_layers)

    def forward(self, x):
        return self.rnn(x)[0]


class CRNN(nn.Module):
    def __init__(self, n_in_channel, nclass, attention=False, activation="Relu", dropout=0,
                 train_cnn=True, rnn_type="BGRU", n_RNN_cell=64, n_layers_RNN=1,
                 dropout_recurrent=0, cnn_integration=False, learned_post=False, **kwargs):
        super().__init__()
        self.cnn = CNN(n_in_channel, activation, dropout, **kwargs)
        self.rnn = BidirectionalGRU(self.cnn.nb_filters[-1], n_RNN_cell, dropout=dropout_recurrent,
                                    num_layers=n_layers_RNN)
        self.dropout = nn.Dropout(dropout)

    def forward(self, x):
        x = self.cnn(x)
        x = x.squeeze(-1).permute(0, 2, 1)
        x = self.dropout(self.rnn(x))
        return x, x


class CRNN_pred(nn.Module):
    """reference src/models/CRNN_GRL.py:206-290: CNN stack, sigmoid on the features themselves, class-softmax attention
    pooling through dense_softmax; the GRU is constructed (state-dict entries) but its call is commented out there."""

    def __init__(self, n_in_channel, nclass, attention=False, activation="Relu", dropout=0, train_cnn=True,
                 rnn_type="BGRU", n_RNN_cell=64, n_layers_RNN=1, dropout_recurrent=0, cnn_integration=False,
                 learned_post=False, **kwargs):
        super().__init__()
        self.dense_softmax = nn.Linear(n_RNN_cell * 2, nclass)
        self.cnn = CNN(n_in_channel, activation, dropout, **kwargs)
        self.rnn = BidirectionalGRU(self.cnn.nb_filters[-1], n_RNN_cell, dropout=dropout_recurrent,
                                    num_layers=n_layers_RNN)

    def forward(self, x, inference=False):
        x = self.cnn(x).squeeze(-1).permute(0, 2, 1)
        strong = torch.sigmoid(x)
        sof = torch.clamp(torch.softmax(self.dense_softmax(x), dim=-1), min=1e-7, max=1)
        weak = (strong * sof).sum(1) / sof.sum(1)
        if inference:
            strong = strong * (weak > 0.5).float().unsqueeze(1)
        return strong, weak


class CNN_FPN(nn.Module):
    """reference src/models/CNN_FPN.py:33-100: the CNN plus two more pyramid levels that REUSE one conv / BatchNorm /
    GLU (cnn_fcn, bn_fcn, glu), each followed by Dropout(0.5) -- a fixed 0.5, not the constructor's dropout -- and
    AvgPool((2,1)).  deconv1/deconv2/conv1x1 exist in the reference module but its forward never calls them; conv1x1
    is kept because it owns state-dict entries."""

    def __init__(self, n_in_channel, activation="Relu", conv_dropout=0, kernel_size=(3, 3, 3), padding=(1, 1, 1),
                 stride=(1, 1, 1), nb_filters=(64, 64, 64), pooling=((1, 4), (1, 4), (1, 4))):
        super().__init__()
        self.nb_filters = list(nb_filters)
        self.cnn = CNN(n_in_channel, activation, conv_dropout, kernel_size, padding, stride, nb_filters, pooling).cnn
        self.cnn_fcn = nn.Conv2d(128, 128, 3, 1, 1)
        self.glu = GLU(128)
        self.pool_fcn = nn.AvgPool2d([2, 1])
        self.bn_fcn = nn.BatchNorm2d(128, eps=0.001, momentum=0.99)
        self.conv1x1 = nn.Conv2d(256, 128, 1)
        self.dropout = nn.Dropout(0.5)

    def level(self, x):
        return self.pool_fcn(self.dropout(self.glu(self.bn_fcn(self.cnn_fcn(x)))))

    def forward(self, x):
        x = self.cnn(x)
        x_2 = self.level(x)
        x_4 = self.level(x_2)
        return x, x_2, x_4


class CRNN_fpn(nn.Module):
    """reference src/models/CRNN_GRL.py:293-389: three BiGRUs on the 313 / 156 / 78 frame levels, coarse levels
    upsampled (bilinear, align_corners=True, hard-coded (156,1) and (313,1)) and fused by 1x1 convolutions"""

    def __init__(self, n_in_channel, nclass, attention=False, activation="Relu", dropout=0, train_cnn=True,
                 rnn_type="BGRU", n_RNN_cell=64, n_layers_RNN=1, dropout_recurrent=0, cnn_integration=False, **kwargs):
        super().__init__()
        self.cnn = CNN_FPN(n_in_channel, activation, dropout, **kwargs)
        nb_in = self.cnn.nb_filters[-1]
        self.rnn = BidirectionalGRU(nb_in, n_RNN_cell, dropout=dropout_recurrent, num_layers=n_layers_RNN)
        self.rnn_2 = BidirectionalGRU(nb_in, n_RNN_cell, dropout=dropout_recurrent, num_layers=n_layers_RNN)
        self.rnn_4 = BidirectionalGRU(nb_in, n_RNN_cell, dropout=dropout_recurrent, num_layers=n_layers_RNN)
        self.dropout = nn.Dropout(dropout)
        self.upsample_2 = nn.Upsample((313, 1), mode="bilinear", align_corners=True)
        self.upsample_4 = nn.Upsample((156, 1), mode="bilinear", align_corners=True)
        self.conv1x1_2 = nn.Conv2d(512, 256, 1)
        self.conv1x1_4 = nn.Conv2d(512, 256, 1)

    def forward(self, x):
        x, x_2, x_4 = self.cnn(x)
        seq = lambda v: v.squeeze(-1).permute(0, 2, 1)               # (B,C,T,1) -> (B,T,C)
        img = lambda v: v.permute(0, 2, 1).unsqueeze(-1)             # (B,T,C) -> (B,C,T,1)
        x = img(self.dropout(self.rnn(seq(x))))
        x_2 = img(self.dropout(self.rnn_2(seq(x_2))))
        x_4 = img(self.dropout(self.rnn_4(seq(x_4))))
        x_2 = self.conv1x1_2(torch.cat((x_2, self.upsample_4(x_4)), 1))
        x = self.conv1x1_4(torch.cat((x, self.upsample_2(x_2)), 1)).squeeze(-1).permute(0, 2, 1)
        return x, x


class Predictor(nn.Module):
    def __init__(self, nclass, attention=False, n_RNN_cell=64, **kwargs):
        super().__init__()
        self.attention = attention
        self.dense = nn.Linear(n_RNN_cell * 2, nclass)
        if attention:
            self.dense_softmax = nn.Linear(n_RNN_cell * 2, nclass)

    def forward(self, x, inference=False):
        strong = torch.sigmoid(self.dense(x))
        if self.attention:
            sof = torch.clamp(torch.softmax(self.dense_softmax(x), dim=-1), min=1e-7, max=1)
            weak = (strong * sof).sum(1) / sof.sum(1)
        else:
            weak = strong.mean(1)
        if inference:
            strong = strong * (weak > 0.5).float().unsqueeze(1)
        return strong, weak


class Clip_Discriminator(nn.Module):
    def __init__(self, input_dim=None, dropout=0):
        super().__init__()
        ch = [1, 128, 64, 32, 16, 8]
        for i in range(5):
            setattr(self, f"conv_{i+1}", nn.Conv2d(ch[i], ch[i + 1], kernel_size=3, stride=2))
        self.dense_d = nn.Linear(16, 1)
        for i in range(5):
            setattr(self, f"bn_{i+1}", nn.BatchNorm2d(ch[i + 1]))

    def forward(self, x):  # (N, T, 256)
        x = x.permute(0, 2, 1).unsqueeze(1)
        for i in range(1, 6):
            x = F.leaky_relu(getattr(self, f"bn_{i}")(getattr(self, f"conv_{i}")(x)), 0.2)
        x = F.adaptive_avg_pool2d(x, (2, 1)).flatten(1)
        return torch.sigmoid(self.dense_d(x))


class Frame_Discriminator(nn.Module):
    """reference src/models/CRNN_GRL.py:116-140: per-frame 256 -> 128 -> 32 -> 1, LeakyReLU(0.2) + Dropout, sigmoid"""

    def __init__(self, input_dim=None, dropout=0):
        super().__init__()
        self.dense_d_1 = nn.Linear(256, 128)
        self.dense_d_2 = nn.Linear(128, 32)
        self.dense_d_3 = nn.Linear(32, 1)
        self.dropout = nn.Dropout(dropout)

    def forward(self, x):
        x = self.dropout(F.leaky_relu(self.dense_d_1(x), 0.2))
        x = self.dropout(F.leaky_relu(self.dense_d_2(x), 0.2))
        return torch.sigmoid(self.dense_d_3(x))


def weights_init(m):
    name = m.__class__.__name__
    if "Conv2d" in name or "Conv1d" in name:
        nn.init.xavier_uniform_(m.weight, gain=np.sqrt(2))
        m.bias.data.fill_(0)
    elif "BatchNorm" in name:
        m.weight.data.normal_(1.0, 0.02)
        m.bias.data.fill_(0)
    elif "GRU" in name:
        for w in m.parameters():
            if w.dim() > 1:
                nn.init.orthogonal_(w.data)
    elif "Linear" in name:
        m.weight.data.normal_(0, 0.01)
        m.bias.data.zero_()


def build(seed=2023, dropout=0.5, crnn_kwargs=None, clip_d=False):
    """Seeded CRNN + Predictor (+ Clip_Discriminator) with the reference's init."""
    torch.manual_seed(seed)
    kw = dict(CRNN_KWARGS if crnn_kwargs is None else crnn_kwargs)
    kw["dropout"] = dropout
    crnn = CRNN(**kw)
    pred = Predictor(**PREDICTOR_KWARGS)
    crnn.apply(weights_init)
    pred.apply(weights_init)
    out = [crnn, pred]
    if clip_d:
        d = Clip_Discriminator()
        d.apply(weights_init)
        out.append(d)
    return out


# ----------------------------------------------------------------------------- schedules / EMA
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


def learning_rate(rampup_value, c_epoch, max_lr, rampdown_value=1.0):
    lr = rampup_value * rampdown_value * max_lr
    if c_epoch > 100:
        lr = lr * (0.5 ** (1 + ((c_epoch - 100) // 20)))
    return lr


def grl_coeff(it, alpha=1.0, lo=0.0, hi=1.0, max_iters=1000.0):
    return float(2.0 * (hi - lo) / (1.0 + np.exp(-alpha * it / max_iters)) - (hi - lo) + lo)


@torch.no_grad()
def update_ema_variables(model, ema_model, alpha, global_step):
    alpha = min(1 - 1 / (global_step + 1), alpha)
    msd, esd = model.state_dict(), ema_model.state_dict()
    for k in esd.keys():
        esd[k] = esd[k].clone() * alpha + msd[k].clone() * (1.0 - alpha)
    try:
        ema_model.load_state_dict(esd)
    except RuntimeError:
        # the reference raises here for a plain CRNN (CNN.state_dict() drops a "cnn." level,
        # DESIGN.md D8); the intended update is applied with the level restored
        ema_model.load_state_dict({("cnn." + k if k.startswith("cnn.") else k): v for k, v in esd.items()})


class _GRL(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, coeff):
        ctx.coeff = coeff
        return x * 1.0

    @staticmethod
    def backward(ctx, g):
        return g.neg() * ctx.coeff, None


def domain_loss(disc, f_s, f_t, coeff):
    """cdan_frame.py:89-119 as actually executed: BCE(D(GRL(cat(f_s,f_t))), [1..,0..])."""
    d = disc(_GRL.apply(torch.cat((f_s, f_t), 0), coeff)).squeeze()
    lab = torch.cat((torch.ones(f_s.size(0)), torch.zeros(f_t.size(0)))).to(d.dtype)
    return F.binary_cross_entropy(d, lab)


def train_losses(crnn, pred, x_syn, y_syn, x_real=None, y_weak_real=None, ema=None, x_real_ema=None,
                 consistency_cost=1.0):
    """Loss of one ``train_mt`` iteration without ISP (main_baseline.py:337-341,431-498).
    Returns (loss, dict of parts / predictions)."""
    bce, mse = nn.BCELoss(), nn.MSELoss()
    enc_s, _ = crnn(x_syn)
    strong_s, weak_s = pred(enc_s)
    out = dict(strong_syn=strong_s, weak_syn=weak_s, enc_syn=enc_s)
    weak_loss = bce(weak_s, y_syn.max(-2)[0])
    strong_loss = bce(strong_s, y_syn)
    loss = strong_loss + weak_loss
    if x_real is not None:
        enc_r, _ = crnn(x_real)
        strong_r, weak_r = pred(enc_r)
        out.update(strong_real=strong_r, weak_real=weak_r, enc_real=enc_r)
        if ema is not None:
            loss = loss + bce(weak_r, y_weak_real)
            with torch.no_grad():
                enc_e, _ = ema[0](x_real_ema)
                strong_e, weak_e = ema[1](enc_e)
            out.update(strong_ema=strong_e, weak_ema=weak_e)
            loss = loss + consistency_cost * mse(weak_r, weak_e) + consistency_cost * mse(strong_r, strong_e)
    out["loss"] = loss
    return loss, out


def _roll_each(x, shifts, dim):
    """per-sample torch.roll, as the reference's python loops do (src/main_baseline.py:234-246, 375-388)"""
    return torch.stack([torch.roll(x[k], int(shifts[k]), dims=dim) for k in range(x.shape[0])], 0)


def train_losses_isp(crnn, pred, ema, x_syn, y_syn, x_real, y_weak_real, x_real_ema, shift_frames, shift_bins,
                     consistency_cost=1.0, pooling_time_ratio=4):
    """Loss of one ``train_mt`` iteration with ``-mt -ISP`` (src/main_baseline.py:229-277, 337-420, 431-529)."""
    bce, mse = nn.BCELoss(), nn.MSELoss()
    cc = consistency_cost
    weak_index = y_weak_real.shape[0] // 2
    # inputs are (B,1,T,F): sample k is (1,T,F) -> time is dim 1, frequency dim 2
    x_real_sh, x_real_fs = _roll_each(x_real, shift_frames, 1), _roll_each(x_real, shift_bins, 2)
    x_ema_sh, x_ema_fs = _roll_each(x_real_ema, shift_frames, 1), _roll_each(x_real_ema, shift_bins, 2)
    x_syn_sh, x_syn_fs = _roll_each(x_syn, shift_frames, 1), _roll_each(x_syn, shift_bins, 2)
    pool_shift = [int(s / pooling_time_ratio) for s in shift_frames]

    enc_s, _ = crnn(x_syn); strong_s, weak_s = pred(enc_s)
    enc_r, _ = crnn(x_real); strong_r, weak_r = pred(enc_r)
    with torch.no_grad():
        se, we = ema[1](ema[0](x_real_ema)[0])
        se_sh, _ = ema[1](ema[0](x_ema_sh)[0])
        se_fs, _ = ema[1](ema[0](x_ema_fs)[0])
    strong_r_roll = _roll_each(strong_r, pool_shift, 0).detach()
    strong_s_roll = _roll_each(strong_s, pool_shift, 0).detach()
    y_syn_roll = _roll_each(y_syn, pool_shift, 0)
    strong_r_sh, weak_r_sh = pred(crnn(x_real_sh)[0])
    strong_r_fs, weak_r_fs = pred(crnn(x_real_fs)[0])
    strong_s_sh, weak_s_sh = pred(crnn(x_syn_sh)[0])
    strong_s_fs, weak_s_fs = pred(crnn(x_syn_fs)[0])

    y_weak_syn = y_syn.max(-2)[0]
    weak_class = bce(weak_s, y_weak_syn) + bce(weak_r, y_weak_real)
    weak_fs_class = bce(weak_s_fs, y_weak_syn) + bce(weak_r_fs[:weak_index], y_weak_real[:weak_index])
    strong_class = bce(strong_s, y_syn)
    strong_sh_class = bce(strong_s_sh, y_syn_roll)
    strong_fs_class = bce(strong_s_fs, y_syn)
    cons_strong = cc * mse(strong_r, se)
    cons_weak = cc * mse(weak_r, we)
    cons_strong_sh = cc * mse(strong_r_sh, se_sh)
    cons_strong_fs = cc * mse(strong_r_fs, se_fs)
    loss = strong_class + weak_class + (cons_weak + cons_strong)
    cons_shift = cc / 2 * (mse(strong_s_sh, strong_s_roll) + mse(strong_r_sh, strong_r_roll))
    loss = loss + (weak_fs_class + strong_sh_class + strong_fs_class + cons_shift)
    loss = loss + 1 / 2 * (cons_strong_sh + cons_strong_fs)
    return loss
