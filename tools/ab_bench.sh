#!/bin/bash
# A/B of whole-step bench lines between library builds on ONE box: tools/ab_bench.sh <mode args...> -- <lib tag> [<lib tag> ...]
#   default build first and last (drift check); results under gpurun_out/ab_<tag>.json
D=$PWD/bird-sound-event-detecion_amd
args=(); while [ $# -gt 0 ] && [ "$1" != "--" ]; do args+=("$1"); shift; done; shift || true
run() { tag=$1; lib=$2; BSED_LIB_PATH=$lib python bench.py --no-cpu-baseline --steps 20 --warmup 3 "${args[@]}" > gpurun_out/ab_$tag.json 2> gpurun_out/ab_$tag.err
  python - <<PY
import json
d=json.loads(open("gpurun_out/ab_$tag.json").read().strip().splitlines()[-1])
print("$tag", d["ms_per_step"], "ms", d["value"], d["unit"])
PY
}
run default $D/libbsed.so
for t in "$@"; do run $t $D/libbsed_$t.so; done
run default2 $D/libbsed.so
