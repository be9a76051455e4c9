"""Checkpoint dictionary of the reference (src/main_baseline.py:895-971,1040-1074): same keys, so files written by
either side load in the other ("model", "model_p", ["model_d"], ["model_ema", "model_p_ema"], "optimizer", ...,
"pooling_time_ratio", "many_hot_encoder", "median_window", "epoch").  Model state dicts are moved to the CPU on save;
the "cnn." / "cnn.cnn." key quirk of reference checkpoints is absorbed by CRNN.load_state_dict.  The "optimizer" /
"optimizer_d" entries hold FlatAdam / FlatSGD state in torch.optim's own format (per-parameter exp_avg / exp_avg_sq /
momentum_buffer in the reference's parameter order, CPU tensors), so ``torch.optim.Adam(...).load_state_dict(
state["optimizer"]["state_dict"])`` of the reference (src/main_baseline.py:874) accepts them and vice versa."""
import torch


def _entry(module, kwargs):
    return {"name": module.__class__.__name__, "args": "", "kwargs": kwargs,
            "state_dict": {k: v.detach().cpu().clone() for k, v in module.state_dict().items()}}


def build_state(crnn, predictor, crnn_kwargs, predictor_kwargs, optimizer=None, many_hot_encoder=None,
                pooling_time_ratio=4, median_window=14, epoch=0, crnn_ema=None, predictor_ema=None,
                discriminator=None, discriminator_kwargs=None, optimizer_d=None):
    state = {"model": _entry(crnn, crnn_kwargs), "model_p": _entry(predictor, predictor_kwargs),
             "pooling_time_ratio": pooling_time_ratio, "median_window": median_window, "epoch": epoch}
    if many_hot_encoder is not None:
        state["many_hot_encoder"] = many_hot_encoder.state_dict()
    if optimizer is not None:
        state["optimizer"] = {"name": optimizer.__class__.__name__, "args": "", "kwargs": {},
                              "state_dict": optimizer.state_dict()}
    if discriminator is not None:
        state["model_d"] = _entry(discriminator, discriminator_kwargs or {})
        if optimizer_d is not None and hasattr(optimizer_d, "state_dict"):
            state["optimizer_d"] = {"name": optimizer_d.__class__.__name__, "args": "", "kwargs": {},
                                    "state_dict": optimizer_d.state_dict()}
    if crnn_ema is not None:
        state["model_ema"] = _entry(crnn_ema, crnn_kwargs)
        state["model_p_ema"] = _entry(predictor_ema, predictor_kwargs)
    return state


def save(state, path):
    torch.save(state, path)


def load_models(path_or_state, device="cuda"):
    """-> dict(crnn, predictor[, crnn_ema, predictor_ema, discriminator]) rebuilt from the stored kwargs"""
    from .disc import Clip_Discriminator
    from .models import CRNN, Predictor
    st = torch.load(path_or_state, map_location="cpu", weights_only=False) if isinstance(path_or_state, str) else path_or_state
    out = {}
    crnn = CRNN(**st["model"]["kwargs"]); crnn.load_state_dict(st["model"]["state_dict"]); out["crnn"] = crnn
    pred = Predictor(**st["model_p"]["kwargs"]); pred.load_state_dict(st["model_p"]["state_dict"]); out["predictor"] = pred
    if "model_ema" in st:
        e = CRNN(**st["model_ema"]["kwargs"]); e.load_state_dict(st["model_ema"]["state_dict"]); out["crnn_ema"] = e
        p = Predictor(**st["model_p_ema"]["kwargs"]); p.load_state_dict(st["model_p_ema"]["state_dict"]); out["predictor_ema"] = p
    if "model_d" in st:
        d = Clip_Discriminator(**st["model_d"]["kwargs"]); d.load_state_dict(st["model_d"]["state_dict"]); out["discriminator"] = d
    out["state"] = st
    return out
