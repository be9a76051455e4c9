"""A/B of the discriminator lowering: im2col everywhere vs the direct (space-to-depth) layers, against the fp64 oracle."""
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from oracle import crnn_oracle as co  # noqa: E402
import bsed_amd.disc as D  # noqa: E402

B, T = int(sys.argv[1]) if len(sys.argv) > 1 else 24, int(sys.argv[2]) if len(sys.argv) > 2 else 216
rng = np.random.default_rng(31)
fn = rng.standard_normal((B, T, 256)).astype(np.float32)
torch.manual_seed(5)
ref = D.Clip_Discriminator()
od = co.Clip_Discriminator().double()
od.load_state_dict({k: (v.double() if v.is_floating_point() else v) for k, v in ref.state_dict().items()})
od.train()
x = torch.from_numpy(fn).double().requires_grad_()
loss_ref = co.domain_loss(od, x[:B // 2], x[B // 2:], 0.37)
loss_ref.backward()
f = torch.from_numpy(fn).cuda()
res = {}
for name, layers, mode in (("im2col/fp32", (), "fp32"), ("direct/fp32", (2, 3), "fp32"), ("direct2/fp32", (2,), "fp32"),
                           ("direct3/fp32", (3,), "fp32"), ("im2col/bf16x3", (), "bf16x3"), ("direct/bf16x3", (2, 3), "bf16x3")):
    D.DIRECT_LAYERS = layers
    disc = D.Clip_Discriminator()
    disc.load_state_dict(ref.state_dict())
    disc.conv_mode = mode
    disc.train(); disc.zero_grad()
    d, ctx = disc.run_forward(f, n_source=B // 2)
    df = disc.run_backward(ctx, 0.37)
    loss = float(ctx["lossp"][:, 0, 0].sum() / B)
    e_df = float((df.cpu().double() - x.grad).norm() / x.grad.norm())
    errs = {}
    for k, p in od.named_parameters():
        if k.startswith("conv_") and k.endswith("bias"):
            continue
        errs[k] = float((disc.P(k).grad.cpu().double() - p.grad).norm() / (p.grad.norm() + 1e-30))
    res[name] = (df.clone(), {k: disc.P(k).grad.clone() for k in errs})
    print(f"{name:14s} loss err {abs(loss - float(loss_ref)):.2e}  df {e_df:.2e}  worst param {max(errs, key=errs.get)} {max(errs.values()):.2e}", flush=True)
a, b = res["im2col/fp32"], res["direct/fp32"]
print("direct vs im2col (fp32): df rel", float((a[0] - b[0]).norm() / a[0].norm()))
diff = (a[0] - b[0]).abs()
print("  max abs diff at", np.unravel_index(int(diff.argmax()), diff.shape), float(diff.max()), "df max", float(a[0].abs().max()))
print("  per-time-row rel err (first/last 4):", [float(diff[:, t].norm() / a[0][:, t].norm()) for t in (0, 1, 2, 3, T - 4, T - 3, T - 2, T - 1)])
print("  per-feature-col rel err (first/last 4):", [float(diff[:, :, c].norm() / a[0][:, :, c].norm()) for c in (0, 1, 2, 3, 252, 253, 254, 255)])

# ---- layer-by-layer: forward outputs (valid extents) of the direct form vs the im2col form, fp32
outs = {}
for name, layers in (("im2col", ()), ("direct", (2, 3)), ("d3", (3,))):
    D.DIRECT_LAYERS = layers
    disc = D.Clip_Discriminator()
    disc.load_state_dict(ref.state_dict())
    disc.conv_mode = "fp32"
    disc.train(); disc.zero_grad()
    d, ctx = disc.run_forward(f, n_source=B // 2)
    outs[name] = [(l["y"][:, :l["Ho"], :l["Wo"], :].clone(), l["scale"].clone(), l["shift"].clone(), l["mean"].clone()) for l in ctx["layers"]]
for k in range(5):
    for other in ("direct", "d3"):
        a, b = outs["im2col"][k], outs[other][k]
        print(f"layer {k+1} {other}: y rel diff {float((a[0]-b[0]).norm()/a[0].norm()):.2e}  scale {float((a[1]-b[1]).abs().max()):.2e} "
              f"shift {float((a[2]-b[2]).abs().max()):.2e} mean {float((a[3]-b[3]).abs().max()):.2e}")
