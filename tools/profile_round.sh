#!/bin/bash
# Round profile on the GPU box: rocprofv3 kernel stats + PMC passes of the headline bench, bench lines of every mode.
#   bash tools/profile_round.sh r02        (run from the repo root through gpurun; results under gpurun_out/<tag>_*)
# the profiler's preloaded tool initialises HIP before python starts: set the hardware-queue count the step's streams
# expect here, not at import (bsed_amd/_lib.py only warns when it is too late)
export GPU_MAX_HW_QUEUES=${GPU_MAX_HW_QUEUES:-8}
set -u
TAG=${1:-r02}
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
BENCH="$R/bench.py --no-cpu-baseline"
rocprofv3 --kernel-trace --stats -d $OUT/${TAG}_stats -o out --output-format csv -- python3 $BENCH --steps 10 --warmup 2 > $OUT/${TAG}_stats.log 2>&1
echo "stats done"
for c in FETCH_SIZE WRITE_SIZE; do
  rocprofv3 --kernel-trace --pmc $c -d $OUT/${TAG}_pmc_$c -o out --output-format csv -- python3 $BENCH --no-kernel-timer --steps 2 --warmup 1 > $OUT/${TAG}_pmc_$c.log 2>&1
  echo "pmc $c done"
done
rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_WAVE_CYCLES SQ_BUSY_CYCLES -d $OUT/${TAG}_pmc_mfma -o out --output-format csv -- python3 $BENCH --no-kernel-timer --steps 2 --warmup 1 > $OUT/${TAG}_pmc_mfma.log 2>&1
echo "pmc mfma done"
cd $R
python bench.py --steps 20 --warmup 3 > $OUT/${TAG}_bench.json 2> $OUT/${TAG}_bench.log; echo "bench crnn done"
python bench.py --steps 10 --no-cpu-baseline --mode mt > $OUT/${TAG}_bench_mt.json 2> $OUT/${TAG}_bench_mt.log
python bench.py --steps 10 --no-cpu-baseline --mode ada > $OUT/${TAG}_bench_ada.json 2> $OUT/${TAG}_bench_ada.log
python bench.py --steps 20 --no-cpu-baseline --mode cnn > $OUT/${TAG}_bench_cnn.json 2> $OUT/${TAG}_bench_cnn.log
python bench.py --steps 10 --no-cpu-baseline --sr 32000 > $OUT/${TAG}_bench_sr32000.json 2> $OUT/${TAG}_bench_sr32000.log
BSED_CONV_MODE=fp32 python bench.py --steps 6 --no-cpu-baseline > $OUT/${TAG}_bench_fp32_mode.json 2> $OUT/${TAG}_bench_fp32_mode.log
python bench.py --steps 20 --no-cpu-baseline --batch 24 > $OUT/${TAG}_bench_batch24.json 2> $OUT/${TAG}_bench_batch24.log
echo "bench lines done"
# round 3: the bf16 throughput mode (bench line + kernel stats), the HIP-graph replay at the reference's batch, the CNN tagger in bf16
python bench.py --steps 20 --no-cpu-baseline --dtype bf16 > $OUT/${TAG}_bench_bf16.json 2> $OUT/${TAG}_bench_bf16.log
python bench.py --steps 20 --no-cpu-baseline --dtype bf16 --mode cnn > $OUT/${TAG}_bench_cnn_bf16.json 2> $OUT/${TAG}_bench_cnn_bf16.log
python bench.py --steps 10 --no-cpu-baseline --dtype bf16 --mode mt > $OUT/${TAG}_bench_mt_bf16.json 2> $OUT/${TAG}_bench_mt_bf16.log
BSED_RNN_OVERLAP=0 python bench.py --steps 40 --no-cpu-baseline --batch 24 --graph > $OUT/${TAG}_bench_batch24_graph.json 2> $OUT/${TAG}_bench_batch24_graph.log
cd /tmp
rocprofv3 --kernel-trace --stats -d $OUT/${TAG}_stats_bf16 -o out --output-format csv -- python3 $BENCH --dtype bf16 --steps 10 --warmup 2 > $OUT/${TAG}_stats_bf16.log 2>&1
cd $R
echo "round-3 extras done"
