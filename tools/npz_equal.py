#!/usr/bin/env python3
"""python tools/npz_equal.py a.npz b.npz [rtol] -- exit 0 iff every array is bit-identical (or, with rtol, every s* array
agrees to rtol * max|a|: builds that split the positions into a different number of partial slabs round differently)."""
import sys

import numpy as np

a, b = np.load(sys.argv[1]), np.load(sys.argv[2])
rtol = float(sys.argv[3]) if len(sys.argv) > 3 else None
bad = 0
for k in a.files:
    if rtol is not None:
        if k.startswith("s"):
            d = np.abs(a[k].astype(np.float64) - b[k].astype(np.float64)).max()
            ok = d <= rtol * np.abs(a[k]).max()
            print(f"{k}: max |a-b| = {d:.3e}  max |a| = {np.abs(a[k]).max():.3e}  {'ok' if ok else 'FAIL'}")
            bad += 0 if ok else 1
        continue
    same = a[k].shape == b[k].shape and np.array_equal(np.atleast_1d(a[k]).view(np.uint8), np.atleast_1d(b[k]).view(np.uint8))
    if not same:
        bad += 1
        d = np.abs(a[k].astype(np.float64) - b[k].astype(np.float64))
        print(f"{k}: DIFFERENT  max |a-b| = {d.max():.3e}  (max |a| = {np.abs(a[k]).max():.3e})")
print("identical" if not bad else f"{bad} arrays differ")
sys.exit(1 if bad else 0)
