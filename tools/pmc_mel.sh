# PMC passes over the stft->mel kernel at the bench shape (256 clips x 10 s at 22.05 kHz): tools/pmc_mel.sh <tag>
# the profiler's preloaded tool initialises HIP before python starts: set the hardware-queue count the step's streams
# expect here, not at import (bsed_amd/_lib.py only warns when it is too late)
export GPU_MAX_HW_QUEUES=${GPU_MAX_HW_QUEUES:-8}
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
tag=${1:-mel}
cd /tmp && export TMPDIR=/tmp
rocprofv3 -L > $R/gpurun_out/${tag}_counters.txt 2>&1 || true
i=0
for set in "GRBM_GUI_ACTIVE SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY" \
           "GRBM_GUI_ACTIVE SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_INST_CYCLES_VMEM SQ_INSTS_VMEM_RD" \
           "GRBM_GUI_ACTIVE TA_TA_BUSY_sum TA_BUSY_avr TCP_PENDING_STALL_CYCLES_sum TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum SQ_INSTS_SALU SQ_ACTIVE_INST_SCA"; do
  i=$((i + 1))
  rocprofv3 --kernel-trace --pmc $set -d $R/gpurun_out/${tag}_pmc$i -o out --output-format csv -- python3 $R/tools/mel_pair_ab.py prof > $R/gpurun_out/${tag}_pmc$i.log 2>&1
  rm -f $R/gpurun_out/${tag}_pmc$i/out_kernel_trace.csv
done
python3 - <<PY
import csv, collections, glob, re
for f in sorted(glob.glob("$R/gpurun_out/${tag}_pmc*/out_counter_collection.csv")):
    acc = collections.defaultdict(collections.Counter); n = collections.Counter()
    for r in csv.DictReader(open(f)):
        k = re.sub(r"\(.*$", "", r["Kernel_Name"])
        if "stft_mel" not in k: continue
        acc[k][r["Counter_Name"]] += float(r["Counter_Value"])
        if r["Counter_Name"] == "GRBM_GUI_ACTIVE": n[k] += 1
    for k, v in acc.items():
        print(k, "launches", n[k])
        for c, x in sorted(v.items()):
            print(f"   {c:40s} {x / max(n[k], 1):16.0f} per launch")
PY
