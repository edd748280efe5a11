#!/bin/bash
cd "$(dirname "$0")/.." && export TMPDIR=/tmp
mkdir -p gpurun_out
timeout 1500 python tools/label_study_x2.py 6 4096 31 32 33 2>&1 | grep -v "amdgpu.ids\|Warning\|warn" > gpurun_out/r06_label_study_x2.txt
timeout 900 python tools/label_study_x2.py 3 8192 41 2>&1 | grep -v "amdgpu.ids\|Warning\|warn" >> gpurun_out/r06_label_study_x2.txt
tail -40 gpurun_out/r06_label_study_x2.txt
