#!/usr/bin/env python3
"""Assembly patcher for hazard hunting (tools/asm_variant.sh): `s_nop N` after every vector ALU instruction (v_*, not
v_mfma) in lines [A, B) counted from the kernel's label.   asm_patch_range.py KERNEL A B [N]"""
import re
import sys

kern, A, B = sys.argv[1], int(sys.argv[2]), int(sys.argv[3])
N = int(sys.argv[4]) if len(sys.argv) > 4 else 0
rel = None
out = []
for line in sys.stdin.read().split("\n"):
    out.append(line)
    m = re.match(r"^(_Z\w+):", line)
    if m:
        rel = 1 if kern in m.group(1) else None     # the label is line 1 (awk '/^label/,/s_endpgm/' numbering)
        continue
    if rel is None:
        continue
    rel += 1
    t = line.strip()
    if A <= rel < B and t.startswith("v_") and not t.startswith("v_mfma"):
        out.append(f"\ts_nop {N}")
print("\n".join(out))
