#!/usr/bin/env python3
"""Phase stamps of igemm3n_kernel (diagnostic build: tools/build_variant.sh stamp igemm3n.hip "-DI3N_STAMP", run with
BSED_LIB_PATH=.../libbsed_stamp.so): where a wave's life goes, in shader cycles, median over the tiles of one launch.
    python tools/conv_stamp.py [wpe] [W]"""
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from bsed_amd import ops  # noqa: E402

wpe = int(sys.argv[1]) if len(sys.argv) > 1 else 3
W = int(sys.argv[2]) if len(sys.argv) > 2 else 8
B, H, C = 256, 216, 128
x = torch.randn(B, H, W, C, device="cuda")
w = torch.randn(9, C, C, device="cuda") * 0.05
wt = ops.pack_weight3(w, 9, C, C, C * C, C, 1)
TH, TW = ops.tile_for(W)
ntiles = B * ((H + TH - 1) // TH) * (W // TW)
st = torch.zeros(ntiles * 4 * 20, device="cuda", dtype=torch.int64)
ops.set_igemm3n_wpe(wpe)
ops.IGEMM3N_WPE["stamp"] = st
for _ in range(3):
    ops.igemm3(x, wt, C, B, H, W, C, ops.TAPS3x3, epilogue=ops.EPI_STATS)
torch.cuda.synchronize()
t = st.cpu().numpy().reshape(ntiles, 4, 20).astype(np.int64)
hw, xcc = t[:, 0, 18], t[:, 0, 19] & 0xf
# gfx9 HW_ID: wave_id [3:0], simd_id [5:4], pipe [7:6], cu_id [11:8], sh_id [12], se_id [15:13]
cu = ((hw >> 8) & 0xf) | (((hw >> 12) & 1) << 4) | (((hw >> 13) & 7) << 5) | (xcc << 8)
first = cu[np.argmin(t[:, 0, 0])]
sel = np.where(cu == first)[0]
sel = sel[np.argsort(t[sel, 0, 0])]
t0 = t[sel, 0, 0].min()
print(f"timeline of one CU (id {first:#x}, {len(sel)} workgroups, {len(set(cu))} distinct CU ids): wave 0 of each workgroup, kilocycles")
print("   tile  simd slot  start  c0-taps  c1  c2  c3  taps-end  end")
for i in sel[:18]:
    r = (t[i, 0, :16] - t0) / 1000.0
    print(f"  {i:5d}   {(hw[i] >> 4) & 3}    {hw[i] & 0xf}   {r[0]:7.1f} {r[3]:7.1f} {r[6]:7.1f} {r[9]:7.1f} {r[12]:7.1f} {r[13]:7.1f} {r[15]:7.1f}")
rt = t[:, :, 17] - t[:, :, 16]                       # 100 MHz ticks
clk = (t[:, :, 15] - t[:, :, 0]) / np.maximum(rt, 1) * 0.1   # GHz
print(f"in-kernel clock: median {np.median(clk):.2f} GHz (p10 {np.percentile(clk, 10):.2f}, p90 {np.percentile(clk, 90):.2f}); "
      f"launch span {(t[:, :, 17].max() - t[:, :, 16].min()) / 100.0:.1f} us")
t = t[:, :, :16]
names = ["setup+issue", "c0 write", "c0 barrier", "c0 taps", "c1 write", "c1 barrier", "c1 taps", "c2 write", "c2 barrier",
         "c2 taps", "c3 write", "c3 barrier", "c3 taps", "epilogue issue", "stores landed"]
d = np.diff(t, axis=2)
print(f"W={W} wpe={wpe}: wave life median {np.median(t[:, :, 15] - t[:, :, 0]):.0f} cycles (p10 {np.percentile(t[:, :, 15] - t[:, :, 0], 10):.0f}, "
      f"p90 {np.percentile(t[:, :, 15] - t[:, :, 0], 90):.0f})")
for i, n in enumerate(names):
    print(f"  {n:16s} median {np.median(d[:, :, i]):8.0f}  mean {d[:, :, i].mean():8.0f}  p90 {np.percentile(d[:, :, i], 90):8.0f}")
wg = t[:, :, 15].max(1) - t[:, :, 0].min(1)
print(f"  workgroup life median {np.median(wg):.0f}; MFMA cycles of a wave: {36 * 24 * 32}")
