"""A/B of the two stft->mel kernels (one frame per wave / two frames per wave): run once per setting of BSED_MEL_PAIR
and compare the saved linear mel.  usage: python tools/mel_pair_ab.py save <file> | cmp <a> <b> | prof"""
import os
import sys
import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def main():
    if sys.argv[1] == "cmp":
        a, b = np.load(sys.argv[2]), np.load(sys.argv[3])
        for k in a.files:
            x, y = a[k], b[k]
            print(k, x.shape, "bit-identical" if np.array_equal(x, y) else
                  f"max abs diff {np.abs(x - y).max():.3e} (max value {np.abs(x).max():.3e})")
        return
    import time
    import torch
    from bsed_amd.features import MelFrontEnd, MelConfig
    if sys.argv[1] == "prof":     # three launches at the bench shape, for rocprofv3
        wav = torch.rand(256, 220500, device="cuda") - 0.5
        fe = MelFrontEnd(MelConfig(sr=22050))
        for _ in range(3):
            fe.linear(wav)
        torch.cuda.synchronize()
        return
    out = {}
    for name, cfg, B, n in (("sr22050", MelConfig(sr=22050), 5, 220500), ("sr32000", MelConfig(), 3, 320000),
                            ("short", MelConfig(sr=22050), 2, 4000), ("odd", MelConfig(sr=22050), 3, 255 * 10 + 17)):
        g = torch.Generator(device="cuda").manual_seed(3)
        wav = (torch.rand(B, n, device="cuda", generator=g) - 0.5) * torch.linspace(0.1, 1.0, B, device="cuda")[:, None]
        fe = MelFrontEnd(cfg)
        lin, cmax, ssq = fe.linear(wav) if hasattr(fe, "linear") else fe.mel_linear(wav)
        out[name] = lin.cpu().numpy(); out[name + "_max"] = cmax.cpu().numpy(); out[name + "_ssq"] = ssq.cpu().numpy()
    np.savez(sys.argv[2], **out)
    wav = torch.rand(256, 220500, device="cuda") - 0.5
    fe = MelFrontEnd(MelConfig(sr=22050))
    f = fe.linear if hasattr(fe, "linear") else fe.mel_linear
    for _ in range(3):
        f(wav)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(20):
        f(wav)
    torch.cuda.synchronize()
    print(f"mel_linear B=256 22.05 kHz: {(time.perf_counter() - t0) / 20 * 1e3:.3f} ms")


main()
