for e in "X=0" "BSED_RNN_OVERLAP=0" "DEBUG_CLR_GRAPH_PACKET_CAPTURE=1" "DEBUG_CLR_GRAPH_PACKET_CAPTURE=0" "DEBUG_HIP_FORCE_GRAPH_QUEUES=1" "DEBUG_HIP_FORCE_GRAPH_QUEUES=1 DEBUG_CLR_GRAPH_PACKET_CAPTURE=1 BSED_RNN_OVERLAP=0" "DEBUG_HIP_GRAPH_BATCH_SIZE=256"; do env $e python bench.py --no-cpu-baseline --steps 40 --warmup 3 --batch 24 --graph > gpurun_out/bg.json 2> gpurun_out/bg.err; python - <<PY
import json
try:
    d=json.loads(open("gpurun_out/bg.json").read().strip().splitlines()[-1])
    print("$e:", d["ms_per_step"], "ms", d["value"], "clips/s")
except Exception as ex:
    print("$e: failed", ex)
PY
done
