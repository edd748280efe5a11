#!/bin/bash
# round 6, call 16: SQ / LDS / TCC counters of the two cfg5 roofline kernels (64 clips x 16 frames), separate --pmc passes
cd "$(dirname "$0")/.." && export TMPDIR=/tmp
bash tools/pmc_conv16.sh pmc16s 64 l1.spatial > /dev/null 2>&1
bash tools/pmc_conv16.sh pmc16t 64 l1.temporal > /dev/null 2>&1
python - <<'PY'
import re
for tag in ("pmc16s", "pmc16t"):
    txt = open(f"gpurun_out/{tag}/pmc_conv16.txt").read()
    cur, vals = None, {}
    for l in txt.splitlines():
        if not l.startswith(" "):
            cur = l.strip(); vals[cur] = {}
        else:
            k, v = l.split()[:2]; vals[cur][k] = float(v)
    print(tag)
    for k, d in vals.items():
        if "SQ_VALU_MFMA_BUSY_CYCLES" in d and "GRBM_GUI_ACTIVE" in d and d["GRBM_GUI_ACTIVE"] > 0:
            print(f"# {k}: MFMA busy {d['SQ_VALU_MFMA_BUSY_CYCLES'] / (1024 * d['GRBM_GUI_ACTIVE'] / 8):.3f} of all SIMD cycles")
PY
