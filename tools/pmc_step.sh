#!/bin/bash
# one PMC pass of the headline bench: per-kernel VALU / MFMA / LDS activity and wave residency
#   bash tools/pmc_step.sh <tag>     (through gpurun, from the repo root; results under gpurun_out/<tag>_pmc_sq)
set -u
TAG=${1:-sq}
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_VALU_MFMA_BUSY_CYCLES SQ_ACTIVE_INST_LDS SQ_WAIT_INST_ANY SQ_WAIT_ANY GRBM_GUI_ACTIVE -d $R/gpurun_out/${TAG}_pmc_sq -o out --output-format csv -- python3 $R/bench.py --no-cpu-baseline --no-kernel-timer --steps 2 --warmup 1 > $R/gpurun_out/${TAG}_pmc_sq.log 2>&1
echo done
