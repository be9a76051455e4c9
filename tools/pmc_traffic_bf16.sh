#!/bin/bash
# FETCH_SIZE / WRITE_SIZE passes of the bf16-mode bench (separate passes, as MI355X_MICROARCH.md prescribes):
#   bash tools/pmc_traffic_bf16.sh <tag>    (through gpurun; then tools/pmc_traffic.py on the two CSVs)
export GPU_MAX_HW_QUEUES=${GPU_MAX_HW_QUEUES:-8}
set -u
TAG=${1:-r03}
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd /tmp && export TMPDIR=/tmp
for c in FETCH_SIZE WRITE_SIZE; do
  rocprofv3 --kernel-trace --pmc $c -d $R/gpurun_out/${TAG}_pmc_bf16_$c -o out --output-format csv -- python3 $R/bench.py --no-cpu-baseline --no-kernel-timer --dtype bf16 --steps 2 --warmup 1 > $R/gpurun_out/${TAG}_pmc_bf16_$c.log 2>&1
  echo "pmc bf16 $c done"
done
