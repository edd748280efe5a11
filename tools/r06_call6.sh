#!/bin/bash
cd "$(dirname "$0")/.." && export TMPDIR=/tmp
mkdir -p gpurun_out
{
for rep in 1 2; do
  for nt in 1 0; do for g in 256 384 512 640 768 896 1024 1280 1536 2048; do
    echo "== VGG NT=$nt grid=$g rep $rep"; SELAVI_SK_NT=$nt python tools/sk_bench.py --iters 300 --grid $g
  done; done
done
for nt in 1 0; do for g in 256 512 768 1024 1536 2048; do
  echo "== Kinetics NT=$nt grid=$g"; SELAVI_SK_NT=$nt python tools/sk_bench.py --N 230976 --K 400 --iters 150 --grid $g
done; done
} 2>&1 | grep -v amdgpu.ids > gpurun_out/r06_sk_grid_scan.txt
timeout 600 python tools/feature_pass_batch.py 64 2>&1 | grep -v amdgpu.ids > gpurun_out/r06_feature_pass_tuned.txt
timeout 900 python -m pytest tests/test_infer32_gpu.py -x -q -m gpu -s -k "folded or cluster" > gpurun_out/r06_infer32_tests3.log 2>&1; echo "infer32 tests rc=$?" >> gpurun_out/r06_infer32_tests3.log
cat gpurun_out/r06_sk_grid_scan.txt | python -c "
import sys, json
for l in sys.stdin:
    l=l.strip()
    if l.startswith('=='): print(l, end='  ')
    elif l.startswith('{'): d=json.loads(l); print('%.1f us  %.3f' % (d['us_per_iter'], d['frac_of_8TBs']))
"
cat gpurun_out/r06_feature_pass_tuned.txt; tail -3 gpurun_out/r06_infer32_tests3.log
