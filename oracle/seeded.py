"""Portable seeded weights / inputs for golden vectors.  TEST INFRASTRUCTURE ONLY.

numpy's PCG64 ``default_rng`` is bit-reproducible across machines, unlike LAPACK-backed
``orthogonal_`` init, so goldens store only (seed, checksum, expected outputs) and both
the generator (oracle/gen_golden.py, run against the reference) and the tests rebuild
identical weights from the seed.  The values are deliberately *not* the reference's
tiny N(0, 0.01) Linear init: larger weights make kernel errors visible.
"""
import numpy as np


def _fans(shape):
    if len(shape) < 2:
        return shape[0], shape[0]
    rf = int(np.prod(shape[2:])) if len(shape) > 2 else 1
    return shape[1] * rf, shape[0] * rf


def seeded_state(shapes, seed):
    """shapes: ordered {name: shape}.  Returns {name: float32/int64 ndarray}."""
    rng = np.random.default_rng(seed)
    out = {}
    for name, shape in shapes.items():
        shape = tuple(shape)
        if name.endswith("num_batches_tracked"):
            out[name] = np.zeros(shape, dtype=np.int64)
        elif name.endswith("running_mean"):
            out[name] = (0.3 * rng.standard_normal(shape)).astype(np.float32)
        elif name.endswith("running_var"):
            out[name] = rng.uniform(0.5, 2.0, shape).astype(np.float32)
        elif ("batchnorm" in name or ".bn_" in name or name.startswith("bn_")) and name.endswith("weight"):
            out[name] = (1.0 + 0.1 * rng.standard_normal(shape)).astype(np.float32)
        elif name.endswith("bias") or "bias_" in name:
            out[name] = rng.uniform(-0.1, 0.1, shape).astype(np.float32)
        else:
            fan_in, _ = _fans(shape)
            a = 1.5 / np.sqrt(fan_in)
            out[name] = rng.uniform(-a, a, shape).astype(np.float32)
    return out


def load_seeded(module, seed):
    """Overwrite every state_dict entry of a torch module with seeded values."""
    import torch
    sd = module.state_dict()
    vals = seeded_state({k: tuple(v.shape) for k, v in sd.items()}, seed)
    tens = {k: torch.from_numpy(v) for k, v in vals.items()}
    try:
        module.load_state_dict(tens)
    except RuntimeError:
        # CNN.state_dict() strips one "cnn." level (reference src/models/CNN.py:71-75); loaders
        # put it back (src/main_baseline.py:829-833)
        module.load_state_dict({("cnn." + k if k.startswith("cnn.") else k): v for k, v in tens.items()})
    return vals


def checksum(vals):
    """Order-independent float64 checksum of a {name: array} dict."""
    return float(sum(np.abs(v.astype(np.float64)).sum() * (1 + (i % 7)) for i, v in enumerate(vals.values())))


def db_like_input(seed, B, T, F=128):
    """Synthetic dB-mel batch (B,1,T,F) float32 in roughly [-80, 0]."""
    rng = np.random.default_rng(seed)
    base = -40.0 + 12.0 * rng.standard_normal((B, 1, T, F))
    ridge = 25.0 * np.exp(-0.5 * ((np.arange(F)[None, :] - rng.uniform(10, F - 10, (B, 1, T, 1))[..., 0][..., None]) / 4.0) ** 2)
    return np.clip(base + ridge, -80.0, 5.0).astype(np.float32)


def strong_targets(seed, B, Tp, C=20):
    rng = np.random.default_rng(seed)
    y = np.zeros((B, Tp, C), dtype=np.float32)
    for b in range(B):
        for _ in range(3):
            c = int(rng.integers(0, C))
            on = int(rng.integers(0, max(1, Tp - 2)))
            off = int(rng.integers(on + 1, Tp + 1))
            y[b, on:off, c] = 1.0
    return y
