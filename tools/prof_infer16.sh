#!/bin/bash
# rocprofv3 kernel trace of the bf16 eval forward (run on the GPU box) -> gpurun_out/infer16_kernel_summary.txt
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
rm -rf /tmp/prof_i16
rocprofv3 --kernel-trace --stats -d /tmp/prof_i16 -o out -- python $R/tools/infer16_bench.py --only16 > /tmp/prof_i16.log 2>&1
grep "clips/s" /tmp/prof_i16.log
python $R/tools/rocprof_summary.py $(find /tmp/prof_i16 -name "*.db" | head -1) 40 > $R/gpurun_out/infer16_kernel_summary.txt 2>&1
head -30 $R/gpurun_out/infer16_kernel_summary.txt
