#!/bin/bash
# PMC passes over the bf16 eval forward (conv_cl16_kernel): where do the cycles go?  Counters in their own runs with
# --kernel-trace only.  Results: gpurun_out/pmc16_summary.txt
cd "$(dirname "$0")/.." && export TMPDIR=/tmp
i=0
for set in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE" \
           "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_WAIT_INST_LDS SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_SALU" \
           "SQ_INSTS_VMEM_RD SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAVES SQ_INST_CYCLES_VMEM SQ_INSTS_MFMA" \
           "FETCH_SIZE" "WRITE_SIZE"; do
  i=$((i+1))
  rocprofv3 --kernel-trace --pmc $set --output-format csv -d gpurun_out/pmc16_$i -o p -- python tools/infer16_bench.py --only16 --batch 16 > gpurun_out/pmc16_$i.log 2>&1 || echo "pass $i failed: $(tail -2 gpurun_out/pmc16_$i.log)"
done
python - > gpurun_out/pmc16_summary.txt <<'PY'
import csv, glob, collections
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob("gpurun_out/pmc16_*/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"]
        if "conv_cl16_kernel" not in k: continue
        acc[(k.replace("void slv::", "")[:40], r.get("Grid_Size", ""))][r["Counter_Name"]].append(float(r["Counter_Value"]))
for (k, grid), cs in sorted(acc.items(), key=lambda kv: -len(kv[1].get("SQ_WAVES", []))):
    if int(grid or 0) < 256 * 1000: continue
    print(k, "grid", grid)
    for c, v in sorted(cs.items()):
        print(f"   {c:28s} mean/dispatch {sum(v) / len(v):16.1f}  n={len(v)}")
PY
head -60 gpurun_out/pmc16_summary.txt
