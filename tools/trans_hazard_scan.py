#!/usr/bin/env python3
"""Scan a gfx950 assembly listing for short distances between a transcendental VALU instruction (v_exp / v_log / v_rcp /
v_rsq / v_sqrt / v_sin / v_cos: the quarter-rate pipe) and the first instruction that READS its result.
    python tools/trans_hazard_scan.py file.s [kernel filter] [max distance]"""
import re
import subprocess
import sys
from collections import Counter

path, flt = sys.argv[1], (sys.argv[2] if len(sys.argv) > 2 else "")
maxd = int(sys.argv[3]) if len(sys.argv) > 3 else 4
TRANS = ("v_exp_f32", "v_log_f32", "v_rcp_f32", "v_rsq_f32", "v_sqrt_f32", "v_sin_f32", "v_cos_f32", "v_rcp_iflag")


def regs(tok):
    tok = tok.strip().lstrip("-|").rstrip("|")
    m = re.fullmatch(r"v\[(\d+):(\d+)\]", tok)
    if m:
        return set(range(int(m.group(1)), int(m.group(2)) + 1))
    m = re.fullmatch(r"v(\d+)", tok)
    return {int(m.group(1))} if m else set()


kernel, insts, res = None, [], {}


def flush():
    global insts
    for i, (op, ops_, ln) in enumerate(insts):
        if not op.startswith(TRANS):
            continue
        d = regs(ops_[0])
        dist = 0
        for j in range(i + 1, min(i + 1 + maxd + 2, len(insts))):
            pop, pops, pln = insts[j]
            if pop == "s_nop":
                dist += int(pops[0]) + 1
                continue
            dist += 1
            srcs = set()
            start = 0 if pop.startswith(("global_store", "scratch_store", "ds_write", "buffer_store")) else 1
            for t in pops[start:]:
                srcs |= regs(t.split(" ")[0])
            if srcs & d:
                if dist <= maxd:
                    res.setdefault(kernel, Counter())[(op.split("_e")[0], pop, dist)] += 1
                break
            if pops and regs(pops[0]) & d and start == 1:
                break   # overwritten
    insts = []


for n, line in enumerate(open(path), 1):
    line = line.split(";")[0].rstrip()
    m = re.match(r"^(_Z\w+):", line)
    if m:
        flush()
        kernel = m.group(1)
        continue
    t = line.strip()
    if not t or t.startswith(".") or kernel is None:
        if t.startswith(".LBB"):
            flush()
        continue
    parts = t.split(None, 1)
    insts.append((parts[0], [x.strip() for x in parts[1].split(",")] if len(parts) > 1 else [], n))
    if parts[0].startswith(("s_cbranch", "s_branch", "s_endpgm")):
        flush()
flush()
names = list(res)
dem = subprocess.run(["c++filt"] + names, capture_output=True, text=True).stdout.splitlines()
for k, d in zip(names, dem):
    if flt in d:
        print(d[:110])
        for (op, pop, dist), c in sorted(res[k].items(), key=lambda x: (x[0][2], x[0][1])):
            print(f"    {op:14s} -> {pop:28s} distance {dist}  x{c}")
