#!/bin/bash
cd "$(dirname "$0")/.." && export TMPDIR=/tmp
mkdir -p gpurun_out
timeout 1800 python -m pytest tests/test_native_comm_gpu.py tests/test_bench_dist_gpu.py -q -m gpu --durations=8 > gpurun_out/r06_rigs.log 2>&1; echo "rigs rc=$?" >> gpurun_out/r06_rigs.log
tail -14 gpurun_out/r06_rigs.log
