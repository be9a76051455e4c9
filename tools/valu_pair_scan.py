#!/usr/bin/env python3
"""Differential hazard hunt: for each assembly listing, the set of (producer opcode, consumer opcode [+ modifiers],
distance) for every register RAW dependency between vector instructions at distance <= D inside one kernel.  Given
listings of builds that FAIL and builds that PASS a bitwise-repeatability check, print the tuples present in every
failing build and in no passing one.
    python tools/valu_pair_scan.py KERNEL D fail1.s fail2.s ... -- pass1.s pass2.s ..."""
import re
import sys

kern, D = sys.argv[1], int(sys.argv[2])
rest = sys.argv[3:]
sep = rest.index("--")
fails, passes = rest[:sep], rest[sep + 1:]


def regs(tok):
    tok = tok.strip().split(" ")[0].lstrip("-|").rstrip("|")
    m = re.fullmatch(r"v\[(\d+):(\d+)\]", tok)
    if m:
        return set(range(int(m.group(1)), int(m.group(2)) + 1))
    m = re.fullmatch(r"v(\d+)", tok)
    return {int(m.group(1))} if m else set()


def scan(path):
    out = set()
    on = False
    insts = []
    for line in open(path):
        line = line.split(";")[0].rstrip()
        m = re.match(r"^(_Z\w+):", line)
        if m:
            on = kern in m.group(1)
            insts = []
            continue
        if not on:
            continue
        t = line.strip()
        if not t or t.startswith("."):
            if t.startswith(".LBB"):
                insts = []
            continue
        parts = t.split(None, 1)
        op = parts[0]
        body = parts[1] if len(parts) > 1 else ""
        mods = " ".join(sorted(set(re.findall(r"(op_sel_hi|op_sel|neg_lo|neg_hi|dst_sel|src0_sel|src1_sel|row_\w+|quad_perm)", body))))
        ops_ = [x.strip() for x in re.split(r",(?![^\[]*\])", body)]
        store = op.startswith(("global_store", "scratch_store", "ds_write", "buffer_store"))
        defs = set() if (store or op.startswith(("s_", "v_cmp"))) else (regs(ops_[0]) if ops_ else set())
        uses = set()
        for tkn in ops_[(0 if store else 1):]:
            uses |= regs(tkn)
        if op.startswith("v_mfma") or op.startswith("v_pk_fma") or op.startswith("v_fma") or op.startswith("v_fmac") or op.startswith("v_mac"):
            pass
        # distance to the producers
        dist = 0
        for pop, pdefs, pmods in reversed(insts[-12:]):
            if pop == "s_nop":
                dist += pdefs + 1
                continue
            dist += 1
            if dist > D:
                break
            if isinstance(pdefs, set) and pdefs & uses:
                out.add((pop.replace("_e32", "").replace("_e64", ""), pmods, op.replace("_e32", "").replace("_e64", ""), mods, dist))
        if op == "s_nop":
            insts.append((op, int(ops_[0]), ""))
        else:
            insts.append((op, defs, mods))
        if op.startswith(("s_cbranch", "s_branch")):
            insts = []
    return out


F = [scan(p) for p in fails]
Pp = [scan(p) for p in passes]
common = set.intersection(*F)
for p in Pp:
    common -= p
print(f"{len(common)} (producer, consumer, distance) tuples in all {len(F)} failing builds and in none of the {len(Pp)} passing builds:")
for t in sorted(common, key=lambda x: (x[4], x[0], x[2])):
    print(f"   {t[0]:22s} [{t[1]:18s}] -> {t[2]:22s} [{t[3]:24s}] distance {t[4]}")
