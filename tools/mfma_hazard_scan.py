#!/usr/bin/env python3
"""Scan a gfx950 assembly listing (hipcc -S --cuda-device-only) for short VALU-write -> MFMA-read distances.

For every v_mfma the nearest preceding instruction IN THE SAME BASIC BLOCK that writes one of its source registers
(A, B or C operand) is found and the issue distance between the two is counted in wait states (one per instruction,
s_nop N counts N + 1).  Printed: per kernel, the histogram of the shortest distances by producer opcode and operand role.
    python tools/mfma_hazard_scan.py file.s [kernel-name-filter] [max-distance-to-list]"""
import re
import sys
from collections import Counter

path = sys.argv[1]
flt = sys.argv[2] if len(sys.argv) > 2 else ""
maxd = int(sys.argv[3]) if len(sys.argv) > 3 else 6


def regs(tok):
    """'v[4:7]' / 'v12' / 'a[0:15]' -> set of (file, index)"""
    m = re.fullmatch(r"([va])\[(\d+):(\d+)\]", tok)
    if m:
        return {(m.group(1), i) for i in range(int(m.group(2)), int(m.group(3)) + 1)}
    m = re.fullmatch(r"([va])(\d+)", tok)
    if m:
        return {(m.group(1), int(m.group(2)))}
    return set()


kernel, block = None, []
res = {}


def flush():
    global block
    for i, (op, ops_) in enumerate(block):
        if not op.startswith("v_mfma"):
            continue
        srcs = [regs(t) for t in ops_[1:4]]
        for role, s in zip("ABC", srcs):
            if not s:
                continue
            dist = 0
            for j in range(i - 1, -1, -1):
                pop, pops = block[j]
                if pop == "s_nop":
                    dist += int(pops[0]) + 1
                    continue
                dist += 1
                if dist > maxd:
                    break
                if pops and regs(pops[0]) & s and not pop.startswith("v_mfma") and not pop.startswith(("ds_", "global_", "buffer_", "scratch_", "s_")):
                    res.setdefault(kernel, Counter())[(pop, role, dist)] += 1
                    break
                if pop.startswith(("ds_read", "global_load", "scratch_load")) and pops and regs(pops[0]) & s:
                    break
    block = []


for line in open(path):
    line = line.split(";")[0].rstrip()
    m = re.match(r"^(_Z\w+):", line)
    if m:
        flush()
        kernel = m.group(1)
        continue
    if re.match(r"^\.?L?BB\d+_\d+:", line.strip()) or line.strip().startswith(".LBB"):
        flush()
        continue
    t = line.strip()
    if not t or t.startswith(".") or kernel is None:
        continue
    parts = t.split(None, 1)
    op = parts[0]
    ops_ = [x.strip() for x in parts[1].split(",")] if len(parts) > 1 else []
    if op.startswith(("s_cbranch", "s_branch", "s_endpgm")):
        block.append((op, ops_))
        flush()
        continue
    block.append((op, ops_))
flush()
import subprocess
names = list(res)
dem = subprocess.run(["c++filt"] + names, capture_output=True, text=True).stdout.splitlines()
for k, d in zip(names, dem):
    if flt in d:
        print(d[:110])
        for (pop, role, dist), n in sorted(res[k].items(), key=lambda x: (x[0][2], x[0][0])):
            print(f"    {pop:28s} -> MFMA Src{role}  distance {dist}  x{n}")
