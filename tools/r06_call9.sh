#!/bin/bash
cd "$(dirname "$0")/.." && export TMPDIR=/tmp
mkdir -p gpurun_out
{
for rep in 1 2; do
  for nt in 0 1; do for g in 256 512 768 1024 1280 1536 2048; do
    echo "== VGG NT=$nt grid=$g rep $rep"; SELAVI_SK_NT=$nt python tools/sk_bench.py --iters 200 --grid $g
  done; done
done
for nt in 0 1; do for g in 256 512 768 1024 1536; do
  echo "== Kinetics NT=$nt grid=$g"; SELAVI_SK_NT=$nt python tools/sk_bench.py --N 230976 --K 400 --iters 100 --grid $g
done; done
for nt in 0 1; do for g in 256 512 768; do echo "== shard 21344 NT=$nt grid=$g"; SELAVI_SK_NT=$nt python tools/sk_bench.py --N 21344 --iters 300 --grid $g; done; done
} 2>&1 | grep -v amdgpu.ids > gpurun_out/r06_sk_ab_honest.txt
cat gpurun_out/r06_sk_ab_honest.txt | python -c "
import sys, json
for l in sys.stdin:
    l=l.strip()
    if l.startswith('=='): print(l, end='  ')
    elif l.startswith('{'): d=json.loads(l); print('%.1f us  %.3f' % (d['us_per_iter'], d['frac_of_8TBs']))
    else: print(l)
"
