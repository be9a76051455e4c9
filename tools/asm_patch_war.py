#!/usr/bin/env python3
"""Assembly patcher for hazard hunting (tools/asm_variant.sh): inserts `s_nop N` after a packed-fp32 vector instruction
(v_pk_*) one of whose SOURCE registers is overwritten by the very next instruction.
    asm_patch_war.py [mode] [N]   mode: all | opsel (only readers that carry op_sel / op_sel_hi) | none (copy)"""
import re
import sys

mode = sys.argv[1] if len(sys.argv) > 1 else "all"
N = int(sys.argv[2]) if len(sys.argv) > 2 else 0


def regs(tok):
    tok = tok.strip().split(" ")[0].lstrip("-|").rstrip("|")
    m = re.fullmatch(r"v\[(\d+):(\d+)\]", tok)
    if m:
        return set(range(int(m.group(1)), int(m.group(2)) + 1))
    m = re.fullmatch(r"v(\d+)", tok)
    return {int(m.group(1))} if m else set()


def parse(line):
    t = line.split(";")[0].strip()
    if not t or t.startswith(".") or t.endswith(":"):
        return None
    parts = t.split(None, 1)
    op, body = parts[0], (parts[1] if len(parts) > 1 else "")
    ops_ = [x.strip() for x in re.split(r",(?![^\[]*\])", body)]
    store = op.startswith(("global_store", "scratch_store", "ds_write", "buffer_store"))
    defs = set() if (store or op.startswith(("s_", "v_cmp"))) else (regs(ops_[0]) if ops_ else set())
    uses = set()
    for tkn in ops_[(0 if store else 1):]:
        uses |= regs(tkn)
    return op, defs, uses, body


lines = sys.stdin.read().split("\n")
out = []
for i, line in enumerate(lines):
    out.append(line)
    if mode == "none":
        continue
    p = parse(line)
    if not p or not p[0].startswith("v_pk_"):
        continue
    if mode == "opsel" and "op_sel" not in p[3]:
        continue
    j = i + 1
    while j < len(lines) and parse(lines[j]) is None:
        j += 1
    if j < len(lines):
        q = parse(lines[j])
        if q and q[1] & p[2]:
            out.append(f"\ts_nop {N}")
print("\n".join(out))
