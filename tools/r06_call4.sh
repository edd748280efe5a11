#!/bin/bash
cd "$(dirname "$0")/.." && export TMPDIR=/tmp
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_sk_gpu.py -q -m gpu > gpurun_out/r06_sk_tests.log 2>&1; echo "sk tests rc=$?" >> gpurun_out/r06_sk_tests.log
timeout 900 python -m pytest tests/test_native_comm_gpu.py -x -q -m gpu -k "sk or sharded" > gpurun_out/r06_sk_comm_tests.log 2>&1; echo "sk comm tests rc=$?" >> gpurun_out/r06_sk_comm_tests.log
timeout 900 python -m pytest tests/test_infer32_gpu.py -x -q -m gpu -s -k "folded or cluster" > gpurun_out/r06_infer32_tests2.log 2>&1; echo "infer32 tests rc=$?" >> gpurun_out/r06_infer32_tests2.log
{
for rep in 1 2; do
  echo "== fused (coherent stores, no fence), rep $rep"; python tools/sk_bench.py --iters 200
  echo "== SELAVI_SK_FUSED=0, rep $rep"; SELAVI_SK_FUSED=0 python tools/sk_bench.py --iters 200
done
echo "== Kinetics size, fused"; python tools/sk_bench.py --N 230976 --K 400 --iters 100 --grid 512
echo "== Kinetics size, unfused"; SELAVI_SK_FUSED=0 python tools/sk_bench.py --N 230976 --K 400 --iters 100 --grid 512
echo "== one shard of 8 (21344 rows), fused / unfused"; python tools/sk_bench.py --N 21344 --iters 500 --grid 512; SELAVI_SK_FUSED=0 python tools/sk_bench.py --N 21344 --iters 500 --grid 512
} > gpurun_out/r06_sk_fused_ab2.txt 2>&1
tail -4 gpurun_out/r06_sk_tests.log; tail -3 gpurun_out/r06_sk_comm_tests.log; grep -n "folded eval\|feature pass\|passed\|failed" gpurun_out/r06_infer32_tests2.log | tail; cat gpurun_out/r06_sk_fused_ab2.txt | grep -v amdgpu.ids
