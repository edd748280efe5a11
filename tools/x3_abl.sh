#!/bin/bash
# timing ablations of igemm3_kernel (forward column only): which part of the chunk loop bounds it
cd "$(dirname "$0")/.."
echo "full"; python tools/conv_bench.py l1.spatial 10 2>&1 | grep l1; python tools/conv_bench.py l2.1.spatial 10 2>&1 | grep l2
for a in 1 2 3 4 5; do
  echo "ABL $a (1 no B loads, 2 no A DMA, 3 no split, 4 no MFMA, 5 no B LDS writes)"
  SELAVI_HIP_LIB=$PWD/tools/proto/libselavi_x3abl$a.so python tools/conv_bench.py l1.spatial 10 2>&1 | grep l1
  SELAVI_HIP_LIB=$PWD/tools/proto/libselavi_x3abl$a.so python tools/conv_bench.py l2.1.spatial 10 2>&1 | grep l2
done
