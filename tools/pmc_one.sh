#!/bin/bash
# FETCH_SIZE / WRITE_SIZE (separate rocprofv3 passes, --kernel-trace only) of the kernels one command launches, mean KB per
# dispatch per (kernel, grid).  Usage: tools/pmc_one.sh <tag> <kernel-substring> -- <command...>
cd "$(dirname "$0")/.." && export TMPDIR=/tmp
tag=$1; sub=$2; shift 3
mkdir -p gpurun_out
for c in FETCH_SIZE WRITE_SIZE; do
  timeout 300 rocprofv3 --kernel-trace --pmc $c --output-format csv -d gpurun_out/pmc1_${tag}_$c -o p -- "$@" > gpurun_out/pmc1_${tag}_$c.log 2>&1
done
python - "$tag" "$sub" <<'PY'
import csv, glob, collections, sys, shutil
tag, sub = sys.argv[1], sys.argv[2]
res = collections.defaultdict(dict)
for c in ("FETCH_SIZE", "WRITE_SIZE"):
    for f in glob.glob(f"gpurun_out/pmc1_{tag}_{c}/**/*counter_collection.csv", recursive=True):
        acc = collections.defaultdict(list)
        for r in csv.DictReader(open(f)):
            if r["Counter_Name"] == c and sub in r["Kernel_Name"]:
                acc[(r["Kernel_Name"].replace("void slv::", "").split("(")[0], r.get("Grid_Size", ""))].append(float(r["Counter_Value"]))
        for k, v in acc.items():
            res[k][c] = sum(v) / len(v); res[k]["n"] = len(v)
    shutil.rmtree(f"gpurun_out/pmc1_{tag}_{c}", ignore_errors=True)
for (k, g), d in sorted(res.items()):
    f, w = d.get("FETCH_SIZE", 0), d.get("WRITE_SIZE", 0)
    print(f"{k} grid={g} n={d['n']}: FETCH_SIZE {f/1024:.1f} MB (x2 = {2*f/1024:.1f}), WRITE_SIZE {w/1024:.1f} MB, hbm = {(2*f+w)/1024:.1f} MB")
PY
