#!/bin/bash
# timing ablations of igemm3_kernel (forward column only): which part of the chunk loop bounds it
cd "$(dirname "$0")/.."
echo "full"; python tools/conv_bench.py l1.spatial 10 2>&1 | grep l1; python tools/conv_bench.py l2.1.spatial 10 2>&1 | grep l2
for a in occ1 abl1 abl2 abl3 abl4; do
  [ -f tools/proto/libselavi_x3$a.so ] || continue
  echo "variant $a (occ1: one workgroup per CU; abl1 no B loads, abl2 no A DMA, abl3 no split, abl4 no MFMA)"
  SELAVI_HIP_LIB=$PWD/tools/proto/libselavi_x3$a.so python tools/conv_bench.py l1.spatial 10 2>&1 | grep l1
  SELAVI_HIP_LIB=$PWD/tools/proto/libselavi_x3$a.so python tools/conv_bench.py l2.1.spatial 10 2>&1 | grep l2
done
