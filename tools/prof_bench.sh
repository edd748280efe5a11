#!/bin/bash
# rocprofv3 kernel trace of the default bench command (run on the GPU box); summary -> gpurun_out/bench_kernel_summary.txt
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
export SELAVI_TUNE_CACHE=/tmp/tune.json
python $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline > /tmp/pre.log 2>&1      # fills the tune cache so the trace holds no timing passes
rm -rf /tmp/prof_bench
rocprofv3 --kernel-trace --stats -d /tmp/prof_bench -o out -- python $R/bench.py --steps 10 --warmup 3 --no-cpu-baseline > $R/gpurun_out/bench_under_rocprof.json 2> /tmp/prof_bench.err
python $R/tools/rocprof_summary.py $(find /tmp/prof_bench -name "*.db" | head -1) 70 > $R/gpurun_out/bench_kernel_summary.txt 2>&1
head -12 $R/gpurun_out/bench_kernel_summary.txt
