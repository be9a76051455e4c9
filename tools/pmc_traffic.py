#!/usr/bin/env python3
"""Turn two rocprofv3 counter-collection CSVs (one `--pmc FETCH_SIZE` pass, one `--pmc WRITE_SIZE` pass of the same
bench command) into profiles/<name>.json: HBM bytes per launch for every kernel.

    python tools/pmc_traffic.py <fetch_counter_collection.csv> <write_counter_collection.csv> <out.json> [note]

FETCH_SIZE / WRITE_SIZE are reported in KiB by rocprofv3; on gfx950 FETCH_SIZE counts a 128-byte request as 64 bytes,
so it is doubled (MI355X_MICROARCH.md, section HBM).  Kernel names are normalised the way bench.py's roofline object
names them (template instance without the argument list)."""
import collections
import csv
import json
import re
import sys


def norm(name):
    name = re.sub(r"^void ", "", name)
    return re.sub(r"\(.*$", "", name)


def per_kernel(path, counter):
    tot, cnt = collections.Counter(), collections.Counter()
    for r in csv.DictReader(open(path)):
        if r["Counter_Name"] != counter:
            continue
        k = norm(r["Kernel_Name"])
        tot[k] += float(r["Counter_Value"])
        cnt[k] += 1
    return tot, cnt


def main():
    fetch_csv, write_csv, out = sys.argv[1:4]
    note = sys.argv[4] if len(sys.argv) > 4 else ""
    ft, fc = per_kernel(fetch_csv, "FETCH_SIZE")
    wt, wc = per_kernel(write_csv, "WRITE_SIZE")
    kernels = {}
    for k in sorted(set(ft) | set(wt)):
        f = 2.0 * 1024.0 * ft[k] / max(fc[k], 1)
        w = 1024.0 * wt[k] / max(wc[k], 1)
        kernels[k] = {"launches_profiled": int(max(fc[k], wc[k])), "fetch_bytes_per_launch_corrected": f,
                      "write_bytes_per_launch": w, "hbm_bytes_per_launch": f + w}
    json.dump({"note": note, "kernels": kernels}, open(out, "w"), indent=1)
    top = sorted(kernels.items(), key=lambda kv: -kv[1]["hbm_bytes_per_launch"] * kv[1]["launches_profiled"])[:12]
    for k, v in top:
        print(f"{k[:60]:60s} x{v['launches_profiled']:<4d} {v['hbm_bytes_per_launch'] / 1e6:10.1f} MB/launch")


if __name__ == "__main__":
    main()
