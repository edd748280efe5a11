#!/bin/bash
# A/B of the split-operand conv kernels (igemm3.hpp) against the native fp32 MFMA kernels: per-op parity, then per-layer timing.
cd "$(dirname "$0")/.."
out=gpurun_out/${1:-x3ab}; mkdir -p $out
python -m pytest tests/test_ops_gpu.py -x -q > $out/ops_test.log 2>&1; tail -3 $out/ops_test.log
SELAVI_CONV_X3=0 python tools/conv_bench.py "" 10 > $out/bench_native.log 2>&1
python tools/conv_bench.py "" 10 > $out/bench_x3.log 2>&1
for v in x3_n2o1 x3_n1o3; do
  [ -f tools/proto/libselavi_$v.so ] && SELAVI_HIP_LIB=$PWD/tools/proto/libselavi_$v.so python tools/conv_bench.py "" 10 > $out/bench_$v.log 2>&1
done
paste -d'\n' $out/bench_native.log $out/bench_x3.log | head -60
