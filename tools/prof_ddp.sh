#!/bin/bash
# kernel-level view of what DDP adds to the SyncBN step (run on the GPU box): two rocprofv3 kernel traces
set -e
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
for mode in syncbn syncbn+ddp-nobcast-view; do
  rm -rf /tmp/prof_$mode
  rocprofv3 --kernel-trace --stats -d /tmp/prof_$mode -o out -- python $R/tools/dist_overhead.py --steps 10 --modes $mode > /tmp/prof_$mode.log 2>&1 || true
  grep "ms/step" /tmp/prof_$mode.log
  python $R/tools/rocprof_summary.py $(find /tmp/prof_$mode -name "*.db" | head -1) > $R/gpurun_out/ddp_prof_$mode.txt 2>&1 || true
  head -3 $R/gpurun_out/ddp_prof_$mode.txt
done
