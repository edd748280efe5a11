#!/bin/bash
cd "$(dirname "$0")/.." && export TMPDIR=/tmp
mkdir -p gpurun_out
timeout 1500 python -m pytest tests/test_infer32_gpu.py tests/test_cluster_gpu.py tests/test_model_gpu.py tests/test_train16_gpu.py -q -m gpu -x > gpurun_out/r06_final_check.log 2>&1; echo "rc=$?" >> gpurun_out/r06_final_check.log
tail -4 gpurun_out/r06_final_check.log
