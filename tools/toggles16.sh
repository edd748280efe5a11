#!/bin/bash
# The cfg5 step (128 clips x 32 frames, 16-bit path) with the step's stream concurrency switched off piece by piece: the
# persistent one-workgroup-per-CU kernels of layer 1 (conv_cl16_sr / _sd / _tr, wgrad_cl16_acc / _tacc) take ~2x their isolated
# time inside the step -- are they waiting for CUs that a concurrent stream's kernels hold?
cd "$(dirname "$0")/.."
for e in "" "SELAVI_OVERLAP_AUDIO=0" "SELAVI_WGRAD_STREAM=0" "SELAVI_WGRAD_STREAM=0 SELAVI_OVERLAP_AUDIO=0"; do
  echo "== $e"; env $e python tools/step16_bench.py ${1:-128} ${2:-32} 5 bf16 2>&1 | tail -1 | cut -c1-220
done
