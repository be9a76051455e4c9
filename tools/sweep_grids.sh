#!/bin/bash
# A/B sweeps of persistent-grid sizes through the env knobs (run through gpurun): prints the matching KERNEL lines
run() { env "$@" python bench.py --no-cpu-baseline --steps 8 2>&1 >/dev/null | grep -E "KERNEL ($PAT)" | sed "s/^.*KERNEL/$* :/"; }
PAT=${PAT:-igemm3s}
for g in 512 768 1024 1536 2048 3072; do run BSED_IGEMM3S_G=$g; done
