#!/bin/bash
# one PMC pass of the headline bench: per-kernel VALU / MFMA / LDS activity and wave residency
#   bash tools/pmc_step.sh <tag>     (through gpurun, from the repo root; results under gpurun_out/<tag>_pmc_sq)
# the profiler's preloaded tool initialises HIP before python starts: set the hardware-queue count the step's streams
# expect here, not at import (bsed_amd/_lib.py only warns when it is too late)
export GPU_MAX_HW_QUEUES=${GPU_MAX_HW_QUEUES:-8}
set -u
TAG=${1:-sq}
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_VALU_MFMA_BUSY_CYCLES SQ_ACTIVE_INST_LDS SQ_WAIT_INST_ANY SQ_WAIT_ANY GRBM_GUI_ACTIVE -d $R/gpurun_out/${TAG}_pmc_sq -o out --output-format csv -- python3 $R/bench.py --no-cpu-baseline --no-kernel-timer --steps 2 --warmup 1 > $R/gpurun_out/${TAG}_pmc_sq.log 2>&1
echo done
