#!/bin/bash
# A/B of two library builds on the GLU kernels: tools/glu_ab.sh <old lib> ["C H W ph pw" ...]
old=$1; shift
for rep in 1 2; do
for cfg in "$@"; do
  BSED_LIB_PATH=$old python tools/glu_time.py $cfg | sed 's/^/old /'
  python tools/glu_time.py $cfg | sed 's/^/new /'
done; done
