#!/bin/bash
# A/B of builds of the library on the 16-bit path in ONE gpurun call: tools/ab16.sh "<variant names>" layers...
cd "$(dirname "$0")/.."
vs=$1; shift
python -m pytest tests/test_ops16_gpu.py tests/test_train16_gpu.py -x -q -k "not sixty and not oracle and not full" 2>&1 | tail -2
for lib in $vs; do
  echo "== tests on $lib"; SELAVI_HIP_LIB=$PWD/tools/proto/libselavi_$lib.so python -m pytest tests/test_ops16_gpu.py tests/test_train16_gpu.py -x -q -k "not sixty and not oracle and not full" 2>&1 | tail -2
done
for rep in 1 2; do
  for lib in base $vs; do
    if [ $lib = base ]; then unset SELAVI_HIP_LIB; else export SELAVI_HIP_LIB=$PWD/tools/proto/libselavi_$lib.so; fi
    echo "== $lib (rep $rep)"
    for L in "$@"; do python tools/conv16_bench.py $L 5 2>&1 | grep "^$L"; done
    python tools/step16_bench.py 128 32 5 bf16 2>&1 | tail -1 | cut -c1-150
  done
done
