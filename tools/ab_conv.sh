#!/bin/bash
# A/B of library builds on conv layers inside ONE gpurun call: tools/ab_conv.sh "<variants>" layers...
cd "$(dirname "$0")/.."
vs=$1; shift
for rep in 1 2; do
  for lib in base $vs; do
    if [ $lib = base ]; then unset SELAVI_HIP_LIB; else export SELAVI_HIP_LIB=$PWD/tools/proto/libselavi_$lib.so; fi
    echo "== $lib (rep $rep)"
    for L in "$@"; do python tools/conv_bench.py $L 10 2>&1 | grep "^$L" | cut -c1-80; done
  done
done
