# the profiler's preloaded tool initialises HIP before python starts: set the hardware-queue count the step's streams
# expect here, not at import (bsed_amd/_lib.py only warns when it is too late)
export GPU_MAX_HW_QUEUES=${GPU_MAX_HW_QUEUES:-8}
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
python3 $R/tools/block0_bench.py 5 > $R/gpurun_out/b0_time.log 2>&1
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_ANY SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_INSTS_VALU SQ_VALU_MFMA_BUSY_CYCLES -d $R/gpurun_out/b0_pmc -o out --output-format csv -- python3 $R/tools/block0_bench.py 1 > $R/gpurun_out/b0_pmc.log 2>&1
rocprofv3 --kernel-trace --pmc GRBM_GUI_ACTIVE SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_WAIT_INST_LDS SQ_ACTIVE_INST_SCA SQ_INST_CYCLES_VMEM -d $R/gpurun_out/b0_pmc2 -o out --output-format csv -- python3 $R/tools/block0_bench.py 1 > $R/gpurun_out/b0_pmc2.log 2>&1
cat $R/gpurun_out/b0_time.log
