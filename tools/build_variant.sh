#!/bin/bash
# A/B builds: tools/build_variant.sh <tag> <file.hip> "<-D flags>"  ->  bird-sound-event-detecion_amd/libbsed_<tag>.so
# (same objects as libbsed.so except <file.hip>, recompiled with the flags; select with BSED_LIB_PATH)
set -euo pipefail
cd "$(dirname "$0")/../bird-sound-event-detecion_amd/csrc"
tag=$1; src=$2; flags=$3
mkdir -p obj/var
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -Wno-unused-result $flags -c "$src" -o "obj/var/${src%.hip}_$tag.o" 2>/dev/null
objs=()
for f in *.hip; do
  if [ "$f" = "$src" ]; then objs+=("obj/var/${src%.hip}_$tag.o"); else objs+=("obj/${f%.hip}.o"); fi
done
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o "../libbsed_$tag.so" "${objs[@]}"
echo "built libbsed_$tag.so"
