"""diagnostic: fused first block vs the four-kernel form, error growth through the network (run on the GPU box)"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import numpy as np, torch
from test_block0_gpu import _pair
from oracle import seeded

B, T, dropout = int(sys.argv[1]), int(sys.argv[2]), float(sys.argv[3])
x = torch.from_numpy(seeded.db_like_input(31, B, T)).cuda()
ocrnn, fused, plain = _pair(dropout, 31, sys.argv[4] if len(sys.argv) > 4 else "bf16x3")
res = []
for m in (fused, plain):
    m.train(); m.set_seed(77)
    enc, ctx = m.run_forward(x, save=True)
    res.append((enc, ctx))
(ef, cf), (ep, cp) = res
for k in ("mean", "invstd", "scale", "shift"):
    a, b = cf["blocks"][0][k], cp["blocks"][0][k]
    print(k, float(((a - b).abs() / b.abs().clamp_min(1e-30)).max()))
for i in range(1, 7):
    a, b = cf["blocks"][i]["inp"], cp["blocks"][i]["inp"]
    print("pooled", i - 1, float((a - b).abs().max()), float(b.abs().max()), float((a - b).norm() / b.norm()))
print("enc", float((ef - ep).abs().max()), float((ef - ep).norm() / ep.norm()))
if dropout == 0:
    ocrnn.train()
    eo, _ = ocrnn(x.cpu())
    print("fused vs oracle", float((ef.cpu() - eo).abs().max()), "plain vs oracle", float((ep.cpu() - eo).abs().max()))
d = torch.from_numpy(np.random.default_rng(5).standard_normal(tuple(ef.shape)).astype(np.float32)).cuda() * 1e-2
for m, c in ((fused, cf), (plain, cp)):
    m.zero_grad(); m.run_backward(c, d)
worst = max(((float((p.grad - plain.P(n).grad).norm() / plain.P(n).grad.norm().clamp_min(1e-30)), n)
             for n, p in fused.named_parameters() if not (".conv" in n and n.endswith(".bias"))))
print("worst grad rel L2", worst)
for n in ("cnn.conv0.weight", "cnn.batchnorm0.weight", "cnn.batchnorm0.bias", "cnn.glu0.linear.weight", "cnn.glu0.linear.bias"):
    print(n, float((fused.P(n).grad - plain.P(n).grad).norm() / plain.P(n).grad.norm()))
