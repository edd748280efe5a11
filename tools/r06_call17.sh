#!/bin/bash
# round 6, call 17: the driver's round-end commands on the final tree: pytest -x -m gpu, smoke(), bench.py
cd "$(dirname "$0")/.." && export TMPDIR=/tmp
mkdir -p gpurun_out
timeout 1800 python -m pytest tests/ -x -q -m gpu --durations=12 > gpurun_out/r06_tests_final2.log 2>&1; echo "tests rc=$?" >> gpurun_out/r06_tests_final2.log
timeout 600 python -c "import __graft_entry__ as g; g.smoke()" > gpurun_out/r06_smoke2.log 2>&1; echo "smoke rc=$?" >> gpurun_out/r06_smoke2.log
timeout 900 python bench.py > gpurun_out/r06_bench_final2.json 2> gpurun_out/r06_bench_final2.err; echo "bench rc=$?"
tail -5 gpurun_out/r06_tests_final2.log; tail -2 gpurun_out/r06_smoke2.log
