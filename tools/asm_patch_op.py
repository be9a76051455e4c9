#!/usr/bin/env python3
"""Assembly patcher for hazard hunting (tools/asm_variant.sh): `s_nop N` after every instruction of kernel KERNEL
whose opcode starts with PREFIX.   asm_patch_op.py KERNEL PREFIX [N]"""
import re
import sys

kern, prefix = sys.argv[1], sys.argv[2]
N = int(sys.argv[3]) if len(sys.argv) > 3 else 0
on = False
out = []
for line in sys.stdin.read().split("\n"):
    out.append(line)
    m = re.match(r"^(_Z\w+):", line)
    if m:
        on = kern in m.group(1)
        continue
    if on and line.strip().startswith(prefix):
        out.append(f"\ts_nop {N}")
print("\n".join(out))
