#!/usr/bin/env python3
"""Regenerate every fixture of tests/golden/ from the reference into a scratch directory and diff it with the
committed files (build container only: needs /root/reference).

    python tools/check_golden.py [case ...]      # exit code 0 = every committed fixture is what the script writes

.npz members are compared bit for bit (dtype, shape, bytes); .json files as parsed values."""
import json
import os
import subprocess
import sys
import tempfile

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
GOLD = os.path.join(ROOT, "tests", "golden")


def diff_npz(a, b):
    A, B = np.load(a, allow_pickle=False), np.load(b, allow_pickle=False)
    bad = sorted(set(A.files) ^ set(B.files))
    for k in sorted(set(A.files) & set(B.files)):
        x, y = A[k], B[k]
        if x.dtype != y.dtype or x.shape != y.shape or x.tobytes() != y.tobytes():
            bad.append(k)
    return bad


def main():
    if not os.path.isdir("/root/reference/src"):
        print("check_golden: /root/reference is not present here; nothing to regenerate")
        return 0
    with tempfile.TemporaryDirectory() as tmp:
        env = dict(os.environ, BSED_GOLDEN_OUT=tmp)
        subprocess.run([sys.executable, os.path.join(ROOT, "oracle", "gen_golden.py")] + sys.argv[1:], check=True, env=env,
                       stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
        made = sorted(os.listdir(tmp))
        failed = 0
        for f in made:
            old, new = os.path.join(GOLD, f), os.path.join(tmp, f)
            if not os.path.exists(old):
                print(f"MISSING  {f}: written by the script, not committed"); failed += 1
                continue
            if f.endswith(".npz"):
                bad = diff_npz(old, new)
            else:
                bad = [] if json.load(open(old)) == json.load(open(new)) else ["<json>"]
            print(f"{'ok      ' if not bad else 'DIFFERS '} {f}" + (f": {bad[:8]}" if bad else ""))
            failed += bool(bad)
        if not sys.argv[1:]:
            for f in sorted(set(os.listdir(GOLD)) - set(made)):
                print(f"ORPHAN   {f}: committed, not written by the script"); failed += 1
        print(f"{len(made) - failed if failed <= len(made) else 0}/{len(made)} fixtures reproduce")
        return 1 if failed else 0


if __name__ == "__main__":
    sys.exit(main())
