#!/bin/bash
# HBM traffic of the two roofline kernels from PMC counters: FETCH_SIZE and WRITE_SIZE in SEPARATE rocprofv3
# passes (--kernel-trace only), averaged per dispatch, FETCH_SIZE doubled per MI355X_MICROARCH.md (gfx950
# reports 1/2 of wide reads).  Writes gpurun_out/pmc_traffic.json; copy it to profiles/r01_pmc.json.
cd "$(dirname "$0")/.." && export TMPDIR=/tmp
mkdir -p gpurun_out
for c in FETCH_SIZE WRITE_SIZE; do
  rocprofv3 --kernel-trace --pmc $c --output-format csv -d gpurun_out/pmct_conv_$c -o p -- python tools/conv_bench.py l1.spatial 3 > gpurun_out/pmct_conv_$c.log 2>&1
  rocprofv3 --kernel-trace --pmc $c --output-format csv -d gpurun_out/pmct_sk_$c -o p -- python tools/sk_bench.py > gpurun_out/pmct_sk_$c.log 2>&1
done
python - <<'PY'
import csv, glob, collections, json
res = collections.defaultdict(dict)
for tag in ("conv", "sk"):
    for c in ("FETCH_SIZE", "WRITE_SIZE"):
        for f in glob.glob(f"gpurun_out/pmct_{tag}_{c}/**/*counter_collection.csv", recursive=True):
            acc = collections.defaultdict(list)
            for r in csv.DictReader(open(f)):
                k = r["Kernel_Name"]
                if r["Counter_Name"] != c or not ("igemm_kernel" in k or "sk_pass_kernel" in k): continue
                acc[(k, r.get("Grid_Size", ""))].append(float(r["Counter_Value"]))
            for (k, grid), v in acc.items():
                name = k.replace("void slv::", "").replace("(slv::IgemmArgs)", "") + " grid=" + grid
                res[name][c + "_KB"] = sum(v) / len(v)
                res[name]["launches"] = len(v)
out = {"_how": "tools/pmc_traffic.sh: rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE in separate passes around "
               "tools/conv_bench.py l1.spatial (B=16 Conv3d 64->144 (1,3,3) fwd/dgrad/wgrad) and tools/sk_bench.py; KB per "
               "dispatch (mean); hbm_bytes = 2*FETCH_SIZE (gfx950 wide-read correction) + WRITE_SIZE"}
for k, d in res.items():
    if "FETCH_SIZE_KB" in d and "WRITE_SIZE_KB" in d:
        d["hbm_bytes_per_launch"] = int((2 * d["FETCH_SIZE_KB"] + d["WRITE_SIZE_KB"]) * 1024)
    out[k] = d
import re
for k, d in list(res.items()):          # aliases bench.py looks up
    m = re.match(r"igemm_kernel<(\d+), (\d+), (\d+), (\w+), (\d+)", k)
    if m and m.group(1) == "0" and m.group(5) == "1":
        out["hot_conv_fwd"] = dict(d, kernel=k)
    if "sk_pass_kernel" in k and k.endswith("grid=262144"):
        out["sk_pass"] = dict(d, kernel=k)
json.dump(out, open("gpurun_out/pmc_traffic.json", "w"), indent=1)
print(json.dumps(out, indent=1))
PY
