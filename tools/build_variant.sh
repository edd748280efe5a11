#!/bin/bash
# Build a variant of the library with extra compiler flags into tools/proto/libselavi_<name>.so (A/B runs on the GPU:
# SELAVI_HIP_LIB=$PWD/tools/proto/libselavi_<name>.so python tools/...).  Usage: tools/build_variant.sh <name> <files...> -- <flags...>
cd "$(dirname "$0")/.."
name=$1; shift
files=(); while [ "$1" != "--" ]; do files+=("$1"); shift; done; shift
mkdir -p /tmp/variant_$name
for f in "${files[@]}"; do
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -Wno-unused-result -mllvm -pragma-unroll-threshold=262144 "$@" -c selavi_amd/csrc/$f -o /tmp/variant_$name/$f.o &
done
wait
objs=$(ls selavi_amd/build/*.o)
for f in "${files[@]}"; do objs=$(echo "$objs" | grep -v "/$f.o"); done
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC $objs /tmp/variant_$name/*.o -ldl -o tools/proto/libselavi_$name.so && echo tools/proto/libselavi_$name.so
