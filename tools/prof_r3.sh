#!/bin/bash
# Round-3 kernel-time profile of the 16-bit step at BASELINE configs[4]'s per-GPU shape (rocprofv3 --kernel-trace --stats, no
# counters).  Untruncated per-(kernel, grid) summary -> gpurun_out/$1/step16_cfg5_kernel_summary.txt (copy to profiles/r03_*).
out=gpurun_out/${1:-prof3}
mkdir -p $out
cd "$(dirname "$0")/.." && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats -d $out/step16_cfg5 -- python tools/step16_bench.py ${2:-128} ${3:-32} 3 bf16 > $out/step16_cfg5.log 2>&1
python tools/rocprof_summary.py $out/step16_cfg5 100000 > $out/step16_cfg5_kernel_summary.txt
rm -rf $out/step16_cfg5
