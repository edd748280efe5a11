#!/bin/bash
cd "$(dirname "$0")/.." && export TMPDIR=/tmp
mkdir -p gpurun_out
for i in 1 2 3; do
  timeout 1500 python -m pytest tests/test_bench_dist_gpu.py -x -q -m gpu -k eight > gpurun_out/r06_eight_$i.log 2>&1; echo "run $i rc=$?"
  grep -n "ILLEGAL\|passed\|failed\|Error" gpurun_out/r06_eight_$i.log | head -5
done
