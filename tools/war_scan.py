#!/usr/bin/env python3
"""WAR scan: vector instructions whose SOURCE register is overwritten by one of the next D instructions.
Prints, per listing, the (reader opcode, writer opcode, distance) tuples for packed-fp32 readers (v_pk_*), and the
differential (in all failing, in no passing build) when '--' separates two groups of listings.
    python tools/war_scan.py KERNEL D fail... -- pass..."""
import re
import sys

kern, D = sys.argv[1], int(sys.argv[2])
rest = sys.argv[3:]
sep = rest.index("--") if "--" in rest else len(rest)
fails, passes = rest[:sep], rest[sep + 1:]


def regs(tok):
    tok = tok.strip().split(" ")[0].lstrip("-|").rstrip("|")
    m = re.fullmatch(r"v\[(\d+):(\d+)\]", tok)
    if m:
        return set(range(int(m.group(1)), int(m.group(2)) + 1))
    m = re.fullmatch(r"v(\d+)", tok)
    return {int(m.group(1))} if m else set()


def scan(path, detail=False):
    out, on, insts, lines = set(), False, [], []
    for ln, line in enumerate(open(path), 1):
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
        op, body = parts[0], (parts[1] if len(parts) > 1 else "")
        ops_ = [x.strip() for x in re.split(r",(?![^\[]*\])", body)]
        store = op.startswith(("global_store", "scratch_store", "ds_write", "buffer_store"))
        defs = set() if (store or op.startswith(("s_", "v_cmp"))) else (regs(ops_[0]) if ops_ else set())
        uses = set()
        for tkn in ops_[(0 if store else 1):]:
            uses |= regs(tkn)
        dist = 0
        for pop, puses, pln in reversed(insts[-8:]):
            if pop == "s_nop":
                dist += puses + 1
                continue
            dist += 1
            if dist > D:
                break
            if isinstance(puses, set) and puses & defs and pop.startswith("v_pk_"):
                key = (pop, op.replace("_e32", "").replace("_e64", ""), dist)
                out.add(key)
                if detail:
                    lines.append((pln, ln, key))
        insts.append((op, int(ops_[0]) if op == "s_nop" else uses, ln))
        if op.startswith(("s_cbranch", "s_branch")):
            insts = []
    return (out, lines) if detail else out


F = [scan(p) for p in fails]
P = [scan(p) for p in passes]
for p, s in zip(fails + passes, F + P):
    print(("FAIL " if p in fails else "PASS ") + p, sorted(s, key=lambda x: x[2]))
if passes:
    common = set.intersection(*F)
    for p in P:
        common -= p
    print("in every failing build and in no passing build:", sorted(common))
