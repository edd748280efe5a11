#!/bin/bash
# round 6, call 1: full GPU suite on the phase-1 tree, the bench line, per-layer 16-bit conv timings at 64 clips
cd "$(dirname "$0")/.." && export TMPDIR=/tmp
mkdir -p gpurun_out
timeout 1500 python -m pytest tests -x -q -m gpu --durations=15 > gpurun_out/r06_tests_a.log 2>&1; echo "tests rc=$?" >> gpurun_out/r06_tests_a.log
timeout 600 python bench.py > gpurun_out/r06_bench_a.json 2> gpurun_out/r06_bench_a.err; echo "bench rc=$?"
timeout 300 python tools/conv16_bench.py "" 5 64 > gpurun_out/r06_conv16_b64.txt 2>&1
tail -5 gpurun_out/r06_tests_a.log; tail -c 600 gpurun_out/r06_bench_a.err
