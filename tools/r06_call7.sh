#!/bin/bash
cd "$(dirname "$0")/.." && export TMPDIR=/tmp
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_sk_gpu.py -q -m gpu > gpurun_out/r06_sk_tests2.log 2>&1; echo "sk tests rc=$?" >> gpurun_out/r06_sk_tests2.log
{
for rep in 1 2; do
  for alt in 0 1; do for nt in 0 1; do for g in 512 768 1280; do
    echo "== VGG ALT=$alt NT=$nt grid=$g rep $rep"; SELAVI_SK_ALT=$alt SELAVI_SK_NT=$nt python tools/sk_bench.py --iters 300 --grid $g
  done; done; done
done
for alt in 0 1; do for nt in 0 1; do for g in 256 512 768; do
  echo "== Kinetics ALT=$alt NT=$nt grid=$g"; SELAVI_SK_ALT=$alt SELAVI_SK_NT=$nt python tools/sk_bench.py --N 230976 --K 400 --iters 150 --grid $g
done; done; done
for alt in 0 1; do echo "== shard 21344 ALT=$alt (forced)"; SELAVI_SK_ALT=$alt python tools/sk_bench.py --N 21344 --iters 500 --grid 512; done
} 2>&1 | grep -v amdgpu.ids > gpurun_out/r06_sk_alt_ab.txt
tail -4 gpurun_out/r06_sk_tests2.log
cat gpurun_out/r06_sk_alt_ab.txt | python -c "
import sys, json
for l in sys.stdin:
    l=l.strip()
    if l.startswith('=='): print(l, end='  ')
    elif l.startswith('{'): d=json.loads(l); print('%.1f us  %.3f' % (d['us_per_iter'], d['frac_of_8TBs']))
"
