#!/bin/bash
cd "$(dirname "$0")/.." && export TMPDIR=/tmp
mkdir -p gpurun_out
{
for rep in 1 2 3; do
  for nt in 0 1; do for rows in 4 8; do
    echo "== NT=$nt ROWS=$rows rep $rep"; SELAVI_SK_NT=$nt SELAVI_SK_ROWS=$rows python tools/sk_bench.py --iters 300 --grid 512
  done; done
done
for nt in 0 1; do for rows in 4 8; do
  echo "== Kinetics NT=$nt ROWS=$rows"; SELAVI_SK_NT=$nt SELAVI_SK_ROWS=$rows python tools/sk_bench.py --N 230976 --K 400 --iters 150 --grid 512
done; done
for nt in 0 1; do echo "== shard 21344 NT=$nt"; SELAVI_SK_NT=$nt python tools/sk_bench.py --N 21344 --iters 500 --grid 512; done
for g in 256 384 768; do echo "== NT=1 grid $g"; SELAVI_SK_NT=1 python tools/sk_bench.py --iters 300 --grid $g; done
} 2>&1 | grep -v amdgpu.ids > gpurun_out/r06_sk_nt_ab.txt
timeout 600 python tools/feature_pass_batch.py 64 128 256 2>&1 | grep -v amdgpu.ids > gpurun_out/r06_feature_pass_batch.txt
cat gpurun_out/r06_sk_nt_ab.txt | python -c "
import sys, json
for l in sys.stdin:
    l=l.strip()
    if l.startswith('=='): print(l, end='  ')
    elif l.startswith('{'): d=json.loads(l); print('%.1f us  %.3f' % (d['us_per_iter'], d['frac_of_8TBs']))
"
cat gpurun_out/r06_feature_pass_batch.txt
