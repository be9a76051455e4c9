#!/usr/bin/env python3
"""Summarise a `rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_WAVE_CYCLES SQ_BUSY_CYCLES` pass of bench.py
into profiles/<name>.json: per kernel, shader cycles per launch (GRBM_GUI_ACTIVE is summed over the 8 XCDs: / 8) and the
fraction of the matrix-core pipe's cycles that were busy (SQ_VALU_MFMA_BUSY_CYCLES counts 32 per 32x32x16 bf16 MFMA and
is summed over the 1024 SIMDs of the chip).

    python tools/pmc_mfma.py <counter_collection.csv> <out.json> [kernel_stats.csv for the matching durations]"""
import collections
import csv
import json
import re
import sys

NSIMD = 256 * 4


def norm(name):
    return re.sub(r"\(.*$", "", re.sub(r"^void ", "", name))


acc, cnt = collections.defaultdict(collections.Counter), collections.Counter()
for r in csv.DictReader(open(sys.argv[1])):
    k = norm(r["Kernel_Name"])
    acc[k][r["Counter_Name"]] += float(r["Counter_Value"])
    if r["Counter_Name"] == "GRBM_GUI_ACTIVE":
        cnt[k] += 1
dur = {}
if len(sys.argv) > 3:
    for r in csv.DictReader(open(sys.argv[3])):
        dur[norm(r["Name"])] = float(r["AverageNs"]) / 1e3
out = {}
for k, v in acc.items():
    if "at::" in k or not cnt[k] or v["GRBM_GUI_ACTIVE"] <= 0:
        continue
    cyc = v["GRBM_GUI_ACTIVE"] / 8.0
    e = {"launches_profiled": cnt[k], "shader_cycles_per_launch": round(cyc / cnt[k]),
         "mfma_pipe_busy_frac": round(v["SQ_VALU_MFMA_BUSY_CYCLES"] / (cyc * NSIMD), 4)}
    if k in dur:
        e["avg_launch_us_unprofiled_run"] = round(dur[k], 1)
        e["effective_clock_ghz"] = round(cyc / cnt[k] / dur[k] / 1e3, 2)
    out[k] = e
json.dump({"source": "rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_WAVE_CYCLES SQ_BUSY_CYCLES -- "
                     "python3 bench.py --no-cpu-baseline --no-kernel-timer --steps 2 --warmup 1",
           "kernels": dict(sorted(out.items(), key=lambda kv: -kv[1]["shader_cycles_per_launch"] * kv[1]["launches_profiled"]))},
          open(sys.argv[2], "w"), indent=1)
for k, e in list(out.items())[:0]:
    print(k, e)
