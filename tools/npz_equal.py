#!/usr/bin/env python3
"""python tools/npz_equal.py a.npz b.npz -- exit 0 iff every array is bit-identical."""
import sys

import numpy as np

a, b = np.load(sys.argv[1]), np.load(sys.argv[2])
bad = 0
for k in a.files:
    same = a[k].shape == b[k].shape and np.array_equal(np.atleast_1d(a[k]).view(np.uint8), np.atleast_1d(b[k]).view(np.uint8))
    if not same:
        bad += 1
        d = np.abs(a[k].astype(np.float64) - b[k].astype(np.float64))
        print(f"{k}: DIFFERENT  max |a-b| = {d.max():.3e}  (max |a| = {np.abs(a[k]).max():.3e})")
print("identical" if not bad else f"{bad} arrays differ")
sys.exit(1 if bad else 0)
