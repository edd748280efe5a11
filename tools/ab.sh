#!/bin/bash
# A/B of builds of the library in ONE gpurun call (boxes differ by +-5 %): tools/ab.sh "<variant names>" layers...
cd "$(dirname "$0")/.."
vs=$1; shift
for rep in 1 2; do
  for lib in base $vs; do
    if [ $lib = base ]; then unset SELAVI_HIP_LIB; else export SELAVI_HIP_LIB=$PWD/tools/proto/libselavi_$lib.so; fi
    echo "== $lib (rep $rep)"
    for L in "$@"; do python tools/conv_bench.py $L 10 2>&1 | grep "^$L"; done
    python tools/step16_bench.py 16 16 20 fp32 2>&1 | tail -1 | cut -c1-110
  done
done
