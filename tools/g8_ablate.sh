#!/bin/bash
# Variants of csrc/conv_cl16_g8.hip for tools/g8_ablate.py (run HERE, before gpurun: the .so files travel with the snapshot)
cd "$(dirname "$0")/.."
specs=("$@"); [ ${#specs[@]} -eq 0 ] && specs=("base:-DSLV_G8_ABL=0" "reqR:-DSLV_G8_REQ_IN_M=0" "noDMA:-DSLV_G8_ABL=1" "noBload:-DSLV_G8_ABL=2" "noLDSw:-DSLV_G8_ABL=3" \
            "noFrag:-DSLV_G8_ABL=4" "noMFMA:-DSLV_G8_ABL=5" "noMbar:-DSLV_G8_ABL=6" "noVMwait:-DSLV_G8_ABL=7" "noEpi:-DSLV_G8_ABL=8" "noLoop:-DSLV_G8_ABL=9" "noStores:-DSLV_G8_ABL=10")
for spec in "${specs[@]}"; do
  tag=${spec%%:*}; flags=${spec#*:}
  tools/build_variant.sh g8_$tag conv_cl16_g8.hip -- $flags -Wno-inline-asm > /dev/null 2>&1 && echo built $tag
done
ls -la tools/proto/libselavi_g8_*.so | wc -l
