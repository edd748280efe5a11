#!/bin/bash
cd "$(dirname "$0")/.." && export TMPDIR=/tmp
mkdir -p gpurun_out
{
echo "== tool, all four grids in one process (the invocation of calls 3-4), new defaults (NT auto)"; python tools/sk_bench.py --iters 200
echo "== tool, default grid, 50 iterations"; python tools/sk_bench.py --iters 50 --grid 768
echo "== tool, default grid, 300 iterations"; python tools/sk_bench.py --iters 300 --grid 768
echo "== tool, old config (NT=0 grid 512), 50 / 300 iterations"; SELAVI_SK_NT=0 python tools/sk_bench.py --iters 50 --grid 512; SELAVI_SK_NT=0 python tools/sk_bench.py --iters 300 --grid 512
} 2>&1 | grep -v amdgpu.ids > gpurun_out/r06_sk_invocation.txt
timeout 600 python bench.py --no-cfg5 --no-native-leg --no-cpu-baseline > gpurun_out/r06_bench_c.json 2> gpurun_out/r06_bench_c.err; echo "bench rc=$?"
SELAVI_SK_NT=0 SELAVI_SK_GRID=512 timeout 600 python bench.py --no-cfg5 --no-native-leg --no-cpu-baseline > gpurun_out/r06_bench_c_oldsk.json 2> gpurun_out/r06_bench_c_oldsk.err; echo "bench rc=$?"
cat gpurun_out/r06_sk_invocation.txt
python - <<'PY'
import json
for f in ("r06_bench_c.json", "r06_bench_c_oldsk.json"):
    d = json.loads(open("gpurun_out/" + f).read().strip().splitlines()[-1])
    print(f, d["value"], d["sk"]["us_per_iter"], d["sk"]["grid"], d["sk"]["roofline"]["frac"], d["sk_round"]["by_feature_pass"])
PY
