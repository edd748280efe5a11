#!/bin/bash
# Round-6 kernel-time profiles (rocprofv3 --kernel-trace --stats, no counters): the bench command (fp32 cfg2 step + SK + the
# 16-bit cfg5 leg) and the 16-bit step at cfg5's per-GPU shape.  Untruncated per-(kernel, grid) summaries ->
# gpurun_out/$1/*_kernel_summary.txt (copy to profiles/r06_*).  The raw traces are removed (gpurun_out is capped at 64 MiB).
out=gpurun_out/${1:-prof6}
mkdir -p $out
cd "$(dirname "$0")/.." && export TMPDIR=/tmp
if [ "$2" != "step-only" ]; then
  # the launch configurations are timed once outside the trace (benchmark mode = cudnn.benchmark, main.py:187) and read back
  # from SELAVI_TUNE_CACHE, so the trace holds the steps themselves and not the tuner's isolated launches
  export SELAVI_TUNE_CACHE=/tmp/selavi_tune_r6.json
  python bench.py --steps 2 --warmup 1 --no-cpu-baseline --cfg5-steps 1 --cfg5-warmup 1 > /dev/null 2>&1
  rocprofv3 --kernel-trace --stats -d $out/bench -- python bench.py --steps 10 --warmup 3 --no-cpu-baseline --cfg5-steps 4 --cfg5-warmup 2 > $out/bench_under_rocprof.json 2> $out/bench.err
  python tools/rocprof_summary.py $out/bench 100000 > $out/bench_kernel_summary.txt
  rm -rf $out/bench
fi
rocprofv3 --kernel-trace --stats -d $out/step16_cfg5 -- python tools/step16_bench.py 128 32 3 bf16 > $out/step16_cfg5.log 2>&1
python tools/rocprof_summary.py $out/step16_cfg5 100000 > $out/step16_cfg5_kernel_summary.txt
rm -rf $out/step16_cfg5
