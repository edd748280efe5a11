#!/bin/bash
# round 6, call 10: kernel-time profiles of the bench command and the cfg5 step (rocprofv3 --kernel-trace --stats), then the
# PMC traffic passes (separate --pmc runs, --kernel-trace only)
cd "$(dirname "$0")/.." && export TMPDIR=/tmp
bash tools/prof_r6.sh prof6
bash tools/pmc_traffic.sh > gpurun_out/pmc_traffic.log 2>&1
ls -la gpurun_out/prof6; head -30 gpurun_out/prof6/bench_kernel_summary.txt; head -12 gpurun_out/prof6/step16_cfg5_kernel_summary.txt; tail -5 gpurun_out/pmc_traffic.log
