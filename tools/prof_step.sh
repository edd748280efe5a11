#!/bin/bash
# rocprofv3 --kernel-trace of one configuration of tools/step16_bench.py -> gpurun_out/$1/step_kernel_summary.txt
# usage: tools/prof_step.sh <tag> <B> <T> <steps> <fp32|bf16>     (SELAVI_WGRAD_STREAM=0 for isolated kernel times)
# The launch configurations are timed once outside the trace and read back from SELAVI_TUNE_CACHE (as tools/prof_r3.sh).
out=gpurun_out/${1:-profstep}
mkdir -p $out
cd "$(dirname "$0")/.." && export TMPDIR=/tmp
export SELAVI_TUNE_CACHE=/tmp/selavi_tune_step.json
python tools/step16_bench.py $2 $3 2 $5 > /dev/null 2>&1
rocprofv3 --kernel-trace --stats -d $out/tr -- python tools/step16_bench.py $2 $3 $4 $5 > $out/step.log 2>&1
python tools/rocprof_summary.py $out/tr 100000 > $out/step_kernel_summary.txt
rm -rf $out/tr
tail -1 $out/step.log | cut -c1-200
