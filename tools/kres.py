#!/usr/bin/env python3
"""Per-kernel resource table of one HIP source: tools/kres.py <file.hip> [name filter] [extra hipcc flags...]
(hipcc -Rpass-analysis=kernel-resource-usage, demangled; compiles to /dev/null)"""
import re
import subprocess
import sys

src = sys.argv[1]
flt = sys.argv[2] if len(sys.argv) > 2 else ""
extra = sys.argv[3:]
cmd = ["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-Wno-unused-result",
       "--cuda-device-only", "-Rpass-analysis=kernel-resource-usage", "-c", src, "-o", "/dev/null"] + extra
err = subprocess.run(cmd, capture_output=True, text=True).stderr
cur, rows = None, []
for line in err.splitlines():
    m = re.search(r"remark: (?:.*?)(Function Name|VGPRs|AGPRs|ScratchSize \[bytes/lane\]|Occupancy \[waves/SIMD\]|SGPRs|VGPRs Spill|LDS Size \[bytes/block\]): (\S+)", line)
    if not m:
        if "error" in line:
            print(line)
        continue
    k, v = m.group(1), m.group(2)
    if k == "Function Name":
        cur = {"name": v}
        rows.append(cur)
    elif cur is not None:
        cur[k] = v
names = subprocess.run(["c++filt"] + [r["name"] for r in rows], capture_output=True, text=True).stdout.splitlines()
for r, n in zip(rows, names):
    if flt in n:
        print(f"{n[:70]:70s} vgpr {r.get('VGPRs'):>4s} agpr {r.get('AGPRs'):>3s} sgpr {r.get('SGPRs'):>3s} scratch {r.get('ScratchSize [bytes/lane]'):>4s} occ {r.get('Occupancy [waves/SIMD]')} lds {r.get('LDS Size [bytes/block]')}")
