#!/bin/bash
# A/B sweeps of persistent-grid sizes through the env knobs (run through gpurun): prints the matching KERNEL lines
run() { env "$@" python bench.py --no-cpu-baseline --steps 8 --timer-steps 0 2>&1 >/dev/null | grep -E "KERNEL ($PAT)" | sed "s/^.*KERNEL/$* :/"; }
PAT=${PAT:-b0_}
for g in 512 1024 2048 4096 8192; do run BSED_B0_FWD_G=$((g*2)) BSED_B0_BWD_G=$g BSED_B0_STATS_G=$g; done
