#!/bin/bash
# round 6, call 11: full GPU suite + smoke + the bench line on the final candidate tree
cd "$(dirname "$0")/.." && export TMPDIR=/tmp
mkdir -p gpurun_out
timeout 1800 python -m pytest tests -q -m gpu --durations=15 > gpurun_out/r06_tests_final.log 2>&1; echo "tests rc=$?" >> gpurun_out/r06_tests_final.log
timeout 600 python -c "import __graft_entry__ as g; g.smoke()" > gpurun_out/r06_smoke.log 2>&1; echo "smoke rc=$?" >> gpurun_out/r06_smoke.log
timeout 900 python bench.py > gpurun_out/r06_bench_final.json 2> gpurun_out/r06_bench_final.err; echo "bench rc=$?"
tail -6 gpurun_out/r06_tests_final.log; tail -2 gpurun_out/r06_smoke.log; tail -c 300 gpurun_out/r06_bench_final.err
