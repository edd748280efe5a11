#!/bin/bash
# Flake check of the 2-rank-on-one-GPU rigs inside ONE gpurun call: tools/flake_cluster.sh <reps> [variant]
cd "$(dirname "$0")/.."
for rep in $(seq 1 $1); do
  for lib in base $2; do
    if [ $lib = base ]; then unset SELAVI_HIP_LIB; else export SELAVI_HIP_LIB=$PWD/tools/proto/libselavi_$lib.so; fi
    echo "== $lib (rep $rep): $(python -m pytest tests/test_cluster_gpu.py -q -k 'syncbn_equals_single or bit_identical' 2>&1 | tail -1)"
  done
done
