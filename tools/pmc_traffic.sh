#!/bin/bash
# HBM traffic of the two roofline kernels from PMC counters: FETCH_SIZE and WRITE_SIZE in SEPARATE rocprofv3
# passes (--kernel-trace only), averaged per dispatch, FETCH_SIZE doubled per MI355X_MICROARCH.md (gfx950
# reports 1/2 of wide reads).  Writes gpurun_out/pmc_traffic.json; copy it to profiles/r04_pmc.json.
cd "$(dirname "$0")/.." && export TMPDIR=/tmp
mkdir -p gpurun_out
for c in FETCH_SIZE WRITE_SIZE; do
  rocprofv3 --kernel-trace --pmc $c --output-format csv -d gpurun_out/pmct_conv_$c -o p -- python tools/conv_bench.py l1.spatial 3 > gpurun_out/pmct_conv_$c.log 2>&1
  rocprofv3 --kernel-trace --pmc $c --output-format csv -d gpurun_out/pmct_sk_$c -o p -- python tools/sk_bench.py > gpurun_out/pmct_sk_$c.log 2>&1
  rocprofv3 --kernel-trace --pmc $c --output-format csv -d gpurun_out/pmct_c16_$c -o p -- python tools/conv16_bench.py l1.spatial 3 > gpurun_out/pmct_c16_$c.log 2>&1
  rocprofv3 --kernel-trace --pmc $c --output-format csv -d gpurun_out/pmct_c16t_$c -o p -- python tools/conv16_bench.py l1.temporal 3 > gpurun_out/pmct_c16t_$c.log 2>&1
done
python - <<'PY'
import csv, glob, collections, json
res = collections.defaultdict(dict)
for tag in ("conv", "sk", "c16", "c16t"):
    for c in ("FETCH_SIZE", "WRITE_SIZE"):
        for f in glob.glob(f"gpurun_out/pmct_{tag}_{c}/**/*counter_collection.csv", recursive=True):
            acc = collections.defaultdict(list)
            for r in csv.DictReader(open(f)):
                k = r["Kernel_Name"]
                if r["Counter_Name"] != c or not ("igemm_kernel" in k or "igemm3_" in k or "sk_pass_kernel" in k or "conv_cl16" in k or "cl16_wgrad" in k): continue
                acc[(k, r.get("Grid_Size", ""))].append(float(r["Counter_Value"]))
            for (k, grid), v in acc.items():
                name = k.replace("void slv::", "").replace("(slv::IgemmArgs)", "").split("(")[0] + " grid=" + grid
                res[name][c + "_KB"] = sum(v) / len(v)
                res[name]["launches"] = len(v)
out = {"_how": "tools/pmc_traffic.sh: rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE in separate passes around "
               "tools/conv_bench.py l1.spatial (B=16 Conv3d 64->144 (1,3,3) fwd/dgrad/wgrad), tools/conv16_bench.py l1.spatial / l1.temporal (the layer-1 convs on the 16-bit path) and tools/sk_bench.py; KB per "
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
    m3 = re.match(r"igemm3_kernel<9, 2, 1, 0", k)          # the split-operand forward of the same layer (csrc/igemm3.hpp)
    if m3:
        out["hot_conv_fwd_x3"] = dict(d, kernel=k)
    if "sk_pass_kernel" in k and k.endswith("grid=262144"):
        out["sk_pass"] = dict(d, kernel=k)
    if k.startswith("conv_cl16_sr_kernel<1, 1>") or ((k.startswith("conv_cl16_s3_kernel<9, 1, 1>") or k.startswith("conv_cl16_kernel<9, 1, 1>")) and "hot_conv16_fwd" not in out):
        out["hot_conv16_fwd"] = dict(d, kernel=k)          # layer-1 spatial train forward of the 16-bit path
    if k.startswith("conv_cl16_tr_kernel<4, 5, 1, 1>"):
        out["hot_conv16_fwd_temporal"] = dict(d, kernel=k)          # layer-1 temporal train forward of the 16-bit path (HBM-bound)
    if k.startswith("cl16_wgrad_acc_kernel<1>") or ((k.startswith("cl16_wgrad3_kernel<5, 1>") or k.startswith("cl16_wgrad_kernel<5, 3, 1>")) and "hot_conv16_wgrad" not in out):
        out["hot_conv16_wgrad"] = dict(d, kernel=k)          # layer-1 spatial weight gradient (accumulator-resident kernel)
json.dump(out, open("gpurun_out/pmc_traffic.json", "w"), indent=1)
import shutil
for tag in ("conv", "sk", "c16", "c16t"):
    for c in ("FETCH_SIZE", "WRITE_SIZE"):
        shutil.rmtree(f"gpurun_out/pmct_{tag}_{c}", ignore_errors=True)
print(json.dumps(out, indent=1))
PY
