"""Time the mel front-end kernels alone (B = 256 clips of 10 s at 22.05 kHz unless told otherwise)."""
import sys

import torch

sys.path.insert(0, __import__("os").path.dirname(__import__("os").path.dirname(__import__("os").path.abspath(__file__))))
from bsed_amd.features import MelConfig, MelFrontEnd  # noqa: E402

B = int(sys.argv[1]) if len(sys.argv) > 1 else 256
sr = int(sys.argv[2]) if len(sys.argv) > 2 else 22050
fe = MelFrontEnd(MelConfig(sr=sr))
wav = torch.randn(B, 10 * sr, device="cuda") * 0.1
for _ in range(3):
    fe.linear(wav)
torch.cuda.synchronize()
n = 20
s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
s.record()
for _ in range(n):
    fe.linear(wav)
e.record()
torch.cuda.synchronize()
ms = s.elapsed_time(e) / n
T = fe.num_frames(wav.shape[1])
gb = 4.0 * B * (wav.shape[1] + T * 128) / 1e9
print(f"mel_linear B={B} sr={sr}: {ms:.3f} ms  ({gb / ms * 1e3:.0f} GB/s algorithmic, {B * T / ms / 1e3:.1f} M frames/s)")
