"""diagnostic: do the split-fp32 GLU kernels give the same bits twice? (shapes of the bench step)"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from bsed_amd import ops
B = int(sys.argv[1]) if len(sys.argv) > 1 else 256
g = torch.Generator(device="cuda").manual_seed(3)
for C, H, W, pool in ((32, 432, 64, (2, 2)), (64, 216, 32, (1, 2)), (128, 216, 16, (1, 2)), (128, 216, 8, (1, 2)),
                      (128, 216, 4, (1, 2)), (128, 216, 2, (1, 2))):
    y = torch.randn(B, H, W, C, device="cuda", generator=g)
    sc = torch.rand(C, device="cuda", generator=g) + 0.5
    sh = torch.randn(C, device="cuda", generator=g) * 0.1
    w = torch.randn(C, C, device="cuda", generator=g) * 0.1
    b = torch.randn(C, device="cuda", generator=g) * 0.1
    dp = torch.randn(B, H // pool[0], W // pool[1], C, device="cuda", generator=g) * 1e-3
    outs = []
    for rep in range(3):
        junk = torch.empty((rep + 1) << 20, device="cuda")
        f = ops.glu_fwd3(y, sc, sh, w, b, B, H, W, C, pool, 0.5, 101, 7)
        if C == 128:
            r = ops.glu_bwd3n(y, sc, sh, w, b, dp, B, H, W, C, pool, 0.5, 101, 7)
        else:
            r = ops.glu_bwd3(y, sc, sh, w, b, dp, B, H, W, C, pool, 0.5, 101, 7)
        torch.cuda.synchronize()
        outs.append([f.clone()] + [t.clone() for t in r if torch.is_tensor(t)])
        del junk
    for rep in (1, 2):
        bad = [i for i, (a, c) in enumerate(zip(outs[0], outs[rep])) if not torch.equal(a, c)]
        print(f"C={C} {H}x{W} rep {rep}: differing outputs {bad}", [float((outs[0][i] - outs[rep][i]).abs().max()) for i in bad])
