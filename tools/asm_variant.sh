#!/bin/bash
# Build libbsed_<tag>.so with ONE source's device code taken from a PATCHED assembly listing (hazard hunting):
#   tools/asm_variant.sh <tag> <file.hip> "<compile flags>" "<python patcher> [args]"
# The patcher is run as:  python3 <patcher> [args] < dev.s > dev_patched.s
set -euo pipefail
tag=$1; src=$2; flags=$3; patcher=$4
R=$(cd "$(dirname "$0")/.." && pwd)
C=$R/bird-sound-event-detecion_amd/csrc
W=$(mktemp -d /tmp/asmvar.XXXX)
LL=/opt/rocm/lib/llvm/bin
cd $C
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -Wno-unused-result $flags --cuda-device-only -S $src -o $W/dev.s 2>/dev/null
(cd $R && python3 $patcher) < $W/dev.s > $W/dev_p.s
$LL/clang -x assembler -target amdgcn-amd-amdhsa -mcpu=gfx950 -c $W/dev_p.s -o $W/dev.o
$LL/ld.lld -shared $W/dev.o -o $W/dev.co
$LL/clang-offload-bundler -type=o -bundle-align=4096 -targets=host-x86_64-unknown-linux-gnu,hipv4-amdgcn-amd-amdhsa--gfx950 -input=/dev/null -input=$W/dev.co -output=$W/dev.hipfb
mkdir -p obj/var
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -Wno-unused-result $flags --cuda-host-only -Xclang -fcuda-include-gpubinary -Xclang $W/dev.hipfb -c $src -o obj/var/${src%.hip}_$tag.o 2>/dev/null
objs=()
for f in *.hip; do
  if [ "$f" = "$src" ]; then objs+=("obj/var/${src%.hip}_$tag.o"); else objs+=("obj/${f%.hip}.o"); fi
done
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o "../libbsed_$tag.so" "${objs[@]}"
echo "built libbsed_$tag.so ($(diff $W/dev.s $W/dev_p.s | grep -c '^>') lines inserted)"
rm -rf $W
