#!/usr/bin/env python3
"""Summarise tools/pmc_step.sh: per kernel (template instance), per launch: shader cycles, the share of them in which
a SIMD's vector ALU / matrix pipe / LDS was busy, resident waves per SIMD, VALU instructions per wave-cycle.
    python tools/pmc_sq.py gpurun_out/<tag>_pmc_sq/out_counter_collection.csv [min_share]"""
import collections, csv, re, sys
NSIMD = 1024
acc = collections.defaultdict(collections.Counter); n = collections.Counter()
for r in csv.DictReader(open(sys.argv[1])):
    k = re.sub(r"\(.*$", "", re.sub(r"^void ", "", r["Kernel_Name"]))
    if "at::" in k or k.startswith("__amd"): continue
    acc[k][r["Counter_Name"]] += float(r["Counter_Value"])
    if r["Counter_Name"] == "GRBM_GUI_ACTIVE": n[k] += 1
rows = []
for k, v in acc.items():
    cyc = v["GRBM_GUI_ACTIVE"] / 8.0          # summed over the 8 XCDs
    if cyc <= 0: continue
    per = cyc / n[k]
    rows.append((cyc, k, n[k], per, 4 * v["SQ_ACTIVE_INST_VALU"] / NSIMD / cyc, v["SQ_VALU_MFMA_BUSY_CYCLES"] / NSIMD / cyc,
                 4 * v["SQ_ACTIVE_INST_LDS"] / NSIMD / cyc * 4, 4 * v["SQ_WAVE_CYCLES"] / NSIMD / cyc,
                 v["SQ_INSTS_VALU"] / NSIMD / cyc, 4 * v["SQ_WAIT_INST_ANY"] / max(4 * v["SQ_WAVE_CYCLES"], 1),
                 4 * v["SQ_WAIT_ANY"] / max(4 * v["SQ_WAVE_CYCLES"], 1)))
tot = sum(r[0] for r in rows)
print(f"{'kernel':46s} {'n':>3s} {'kcyc/launch':>11s} {'share':>6s} {'VALU':>5s} {'MFMA':>5s} {'LDS*':>5s} {'waves':>5s} {'valu/cyc':>8s} {'w_inst':>6s} {'w_any':>6s}")
for r in sorted(rows, reverse=True):
    if r[0] / tot < (float(sys.argv[2]) if len(sys.argv) > 2 else 0.004): continue
    print(f"{r[1][:46]:46s} {r[2]:3d} {r[3] / 1e3:11.1f} {r[0] / tot:6.3f} {r[4]:5.2f} {r[5]:5.2f} {r[6]:5.2f} {r[7]:5.2f} {r[8]:8.3f} {r[9]:6.2f} {r[10]:6.2f}")
