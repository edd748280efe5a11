#!/bin/bash
# round 6, call 2: the folded / two-piece eval forward: op tests, model tests, cluster rounds; the bench line with the new sk_round
cd "$(dirname "$0")/.." && export TMPDIR=/tmp
mkdir -p gpurun_out
timeout 1200 python -m pytest tests/test_infer32_gpu.py -x -q -m gpu -s > gpurun_out/r06_infer32_tests.log 2>&1; echo "infer32 tests rc=$?" >> gpurun_out/r06_infer32_tests.log
timeout 900 python -m pytest tests/test_cluster_gpu.py tests/test_eval_gpu.py tests/test_example_gpu.py -x -q -m gpu > gpurun_out/r06_cluster_tests.log 2>&1; echo "cluster tests rc=$?" >> gpurun_out/r06_cluster_tests.log
timeout 600 python bench.py --no-cfg5 --no-native-leg --no-cpu-baseline > gpurun_out/r06_bench_b.json 2> gpurun_out/r06_bench_b.err; echo "bench rc=$?"
tail -15 gpurun_out/r06_infer32_tests.log; tail -5 gpurun_out/r06_cluster_tests.log; tail -c 400 gpurun_out/r06_bench_b.err
