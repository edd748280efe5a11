#!/bin/bash
# SQ / LDS / TCP counters of the bf16 conv kernels on the layer-1 spatial conv (tools/conv16_bench.py l1.spatial), one
# rocprofv3 --pmc pass per counter group (kernel trace only; each pass under its own timeout: a pass with TA_* / TCP_*STALL*
# counters hung the profiler for the whole gpurun limit once).  Output: gpurun_out/$1/pmc_conv16.txt
out=gpurun_out/${1:-pmc16}
mkdir -p $out
cd "$(dirname "$0")/.." && export TMPDIR=/tmp
i=0
for grp in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_INST_LDS SQ_INSTS_VALU SQ_INSTS_MFMA" \
           "SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_LDS SQ_INSTS_SALU SQ_INSTS_SMEM SQ_VALU_MFMA_BUSY_CYCLES SQ_ACTIVE_INST_VALU" \
           "GRBM_GUI_ACTIVE SQ_WAVES SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_ACTIVE_INST_VMEM SQ_LDS_UNALIGNED_STALL SQ_LDS_ADDR_CONFLICT SQ_INST_LEVEL_LDS" \
           "TCP_TCC_READ_REQ_sum TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum"; do
  i=$((i+1))
  timeout 300 rocprofv3 --kernel-trace --pmc $grp --output-format csv -d $out/p$i -o p -- python tools/conv16_bench.py ${3:-l1.spatial} 3 ${2:-16} > $out/p$i.log 2>&1
done
python - "$out" <<'PY'
import csv, glob, collections, sys
out = sys.argv[1]
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob(f"{out}/p*/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"]
        if "conv_cl16" not in k and "cl16_wgrad" not in k: continue
        name = k.replace("void slv::", "").split("(")[0] + " grid=" + r.get("Grid_Size", "")
        acc[name][r["Counter_Name"]].append(float(r["Counter_Value"]))
with open(f"{out}/pmc_conv16.txt", "w") as fo:
    for name, d in acc.items():
        fo.write(name + "\n")
        for c, v in sorted(d.items()):
            fo.write(f"   {c:28s} {sum(v)/len(v):16.1f}  (n={len(v)})\n")
print(open(f"{out}/pmc_conv16.txt").read())
PY
rm -rf $out/p?
