#!/bin/bash
# Timing ablations of conv_cl16_s3_kernel (csrc/conv_cl16_s3.hip: SLV_S3_ABL): one library per variant, built from the
# objects of the normal build with only that file recompiled; run on the GPU: tools/conv16_bench.py l1.spatial per variant.
cd "$(dirname "$0")/.."
mkdir -p gpurun_out/abl
if [ "$1" == "trace" ]; then
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -Wno-unused-result -DSLV_S3_TRACE -DSLV_S3_TRACE_T0=${2:-0} -c selavi_amd/csrc/conv_cl16_s3.hip -o /tmp/s3_trace.o
  objs=$(ls selavi_amd/build/*.o | grep -v conv_cl16_s3)
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC $objs /tmp/s3_trace.o -ldl -o tools/proto/libselavi_trace.so
  exit 0
fi
if [ "$1" == "build" ]; then
  for v in 1 2 3 4 5 6 7 8; do
    /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -Wno-unused-result -DSLV_S3_ABL=$v -c selavi_amd/csrc/conv_cl16_s3.hip -o /tmp/s3_abl$v.o &
  done
  wait
  objs=$(ls selavi_amd/build/*.o | grep -v conv_cl16_s3)
  for v in 1 2 3 4 5 6 7 8; do
    /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC $objs /tmp/s3_abl$v.o -ldl -o tools/proto/libselavi_abl$v.so
  done
  exit 0
fi
for v in 0 7 8 1 3; do
  lib=selavi_amd/libselavi_hip.so; [ $v != 0 ] && lib=tools/proto/libselavi_abl$v.so
  echo "== ablation $v" ; SELAVI_HIP_LIB=$PWD/$lib python tools/conv16_bench.py l1.spatial 5 ${1:-16} 2>&1 | grep "l1.spatial"
done
