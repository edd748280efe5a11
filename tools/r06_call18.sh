#!/bin/bash
# round 6, call 18: the cfg5 step with the stream concurrency switched off (weight gradients and the audio trunk on the main
# stream): per-kernel durations WITHOUT the stretch of sharing the chip -> what "sum of isolated kernel times" is in the step
cd "$(dirname "$0")/.." && export TMPDIR=/tmp
out=gpurun_out/prof6s; mkdir -p $out
SELAVI_WGRAD_STREAM=0 SELAVI_OVERLAP_AUDIO=0 rocprofv3 --kernel-trace --stats -d $out/serial -- python tools/step16_bench.py 128 32 3 bf16 > $out/serial.log 2>&1
python tools/rocprof_summary.py $out/serial 100000 > $out/step16_cfg5_serial_kernel_summary.txt
rm -rf $out/serial
grep -v "^W2026\|amdgpu.ids" $out/serial.log | tail -5
head -30 $out/step16_cfg5_serial_kernel_summary.txt
