#!/usr/bin/env python3
"""Check the hand-waited prefetch loads of csrc/igemm3.hip (igemm3s_kernel) on the compiler's listing.

The kernel issues its patch prefetch by inline assembly (`global_load_dwordx4` inside ;;#ASMSTART ... ;;#ASMEND) and waits for it
by hand (`s_waitcnt vmcnt(N)` inside ;;#ASMSTART): the compiler does not know the destination registers are in flight, so
NOTHING may read or write them between a load and the next hand-written wait -- a register copy the allocator slips
in there would copy stale data.  For every kernel of the listing this walks every path from an inline-asm load until it
meets the counted inline-asm wait, along fall-through and branch edges, and reports (1) any instruction touching a pending
register and (2) any path on which the counted wait `vmcnt(N)` is reached with fewer than N younger vector-memory
instructions and without the full `vmcnt(0)` wait of the masked-store path in front of it (the wait would then prove
nothing: a masked store, or a spill the compiler added, changes the count).

    hipcc --offload-arch=gfx950 -O3 -std=c++17 --cuda-device-only -S csrc/igemm3.hip -o /tmp/i3.s
    python tools/asm_load_check.py /tmp/i3.s [kernel-name-substring]"""
import re
import sys


def regs_of(tok):
    out = set()
    for m in re.finditer(r"\bv\[(\d+):(\d+)\]|\bv(\d+)\b", tok):
        if m.group(1):
            out.update(range(int(m.group(1)), int(m.group(2)) + 1))
        else:
            out.add(int(m.group(3)))
    return out


def check(name, lines):
    # instruction list with asm-block marks
    ins, labels, in_app = [], {}, False
    for ln in lines:
        t = ln.strip()
        if t.startswith(";;#ASMSTART"):
            in_app = True; continue
        if t.startswith(";;#ASMEND"):
            in_app = False; continue
        m = re.match(r"^(\.LBB\w+):", t)
        if m:
            labels[m.group(1)] = len(ins); continue
        if not t or t.startswith(";") or t.startswith("."):
            continue
        ins.append((t.split(";")[0].strip(), in_app))
    starts = [i for i, (t, a) in enumerate(ins) if a and t.startswith("global_load_dwordx4")]
    bad, seen = [], set()
    # state: (next instruction, pending registers, younger vector-memory instructions (capped), passed a hand vmcnt(0))
    work = [(i + 1, frozenset(regs_of(ins[i][0].split(",")[0])), 0, False) for i in starts]
    while work:
        i, pend, young, drained = work.pop()
        while i < len(ins):
            if (i, pend, young, drained) in seen:
                break
            seen.add((i, pend, young, drained))
            t, a = ins[i]
            op = t.split()[0]
            if a and op == "s_waitcnt" and "vmcnt" in t:
                n = int(re.search(r"vmcnt\((\d+)\)", t).group(1))
                if n == 0:
                    break                               # everything has landed: path ends
                else:
                    # the counted wait: it only proves the load done if at least n vector-memory instructions are
                    # younger than it (fewer -- a masked store path, a spill the compiler added -- and it proves nothing)
                    if not drained and young < n:
                        bad.append((i, f"{t}   <- only {young} younger vector-memory instructions on this path"))
                    break
                i += 1
                continue
            if re.match(r"(global|scratch|buffer|flat)_", op):
                young = min(young + 1, 64)
            if a and op == "global_load_dwordx4":
                pend = pend | frozenset(regs_of(t.split(",")[0]))
            elif regs_of(t) & pend:
                bad.append((i, t))
            if op == "s_endpgm":
                break
            if op.startswith("s_cbranch") or op == "s_branch":
                tgt = t.split()[-1]
                if tgt in labels:
                    work.append((labels[tgt], pend, young, drained))
                if op == "s_branch":
                    break
            i += 1
    return len(starts), sorted(set(bad))


txt = open(sys.argv[1]).read().split("\n")
flt = sys.argv[2] if len(sys.argv) > 2 else "igemm3s_kernel"
cur, body, total_bad, kernels = None, [], 0, 0
for ln in txt + ["_Zend:"]:
    m = re.match(r"^(_Z\w+):", ln)
    if m:
        if cur and flt in cur and body:
            n, bad = check(cur, body)
            if n:
                kernels += 1
                total_bad += len(bad)
                print(f"{cur[:60]:60s} {n:3d} inline loads, {len(bad)} instructions touching a pending register")
                for i, t in bad[:6]:
                    print("      ", t)
        cur, body = m.group(1), []
    elif cur:
        if ln.strip().startswith(".size"):
            if flt in cur and body:
                n, bad = check(cur, body)
                if n:
                    kernels += 1
                    total_bad += len(bad)
                    print(f"{cur[:60]:60s} {n:3d} inline loads, {len(bad)} instructions touching a pending register")
                    for i, t in bad[:6]:
                        print("      ", t)
            cur, body = None, []
        else:
            body.append(ln)
print(f"{kernels} kernels with inline loads, {total_bad} violations")
sys.exit(1 if total_bad else 0)
