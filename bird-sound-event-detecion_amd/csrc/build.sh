#!/bin/bash
# Build libbsed.so for gfx950 in-tree (the .so travels to the GPU box with the snapshot).
set -euo pipefail
cd "$(dirname "$0")"
HIPCC=${HIPCC:-/opt/rocm/bin/hipcc}
FLAGS="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -Wno-unused-result"
mkdir -p obj
pids=()
objs=()
for f in *.hip; do
  o=obj/${f%.hip}.o
  objs+=("$o")
  if [ ! -f "$o" ] || [ "$f" -nt "$o" ] || [ bsed_common.h -nt "$o" ] || [ ../../include/bsed.h -nt "$o" ] \
     || { [ -f igemm_core.h ] && [ igemm_core.h -nt "$o" ]; }; then
    # mel.hip: the SLP vectorizer packs the FFT's scalar fp32 arithmetic into v_pk_* with op_sel swizzles, whose
    # destination-forwarding hazards cost ~90 s_nop per frame: 0.635 -> 0.599 ms without it (A/B on MI355X).
    # The GCN register-pressure trackers let the scheduler see stft_mel2_kernel's real pressure: without them the
    # default max-occupancy scheduler fills all 256 registers and the allocator spills 16-49 dwords (their reloads sit
    # behind the prefetched global loads in vmcnt order: 0.655 ms against 0.603 ms spill-free)
    extra=""; [ "$f" = "mel.hip" ] && extra="-fno-slp-vectorize -mllvm -amdgpu-use-amdgpu-trackers=1"
    $HIPCC $FLAGS $extra -c "$f" -o "$o" &
    pids+=($!)
  fi
done
for p in "${pids[@]:-}"; do [ -n "$p" ] && wait "$p"; done
$HIPCC --offload-arch=gfx950 -shared -fPIC -o ../libbsed.so "${objs[@]}"
echo "built $(realpath ../libbsed.so)"
