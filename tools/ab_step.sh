#!/bin/bash
# A/B of library builds / environment settings on the fp32 step inside ONE gpurun call: tools/ab_step.sh "<variants>" "<env settings>"
cd "$(dirname "$0")/.."
for rep in 1 2 3; do
  for lib in base $1; do
    if [ $lib = base ]; then unset SELAVI_HIP_LIB; else export SELAVI_HIP_LIB=$PWD/tools/proto/libselavi_$lib.so; fi
    echo "== $lib (rep $rep): $(python tools/step16_bench.py 16 16 20 fp32 2>&1 | tail -1 | cut -c1-60)"
  done
  unset SELAVI_HIP_LIB
  for e in $2; do
    echo "== env $e (rep $rep): $(env $e python tools/step16_bench.py 16 16 20 fp32 2>&1 | tail -1 | cut -c1-60)"
  done
done
