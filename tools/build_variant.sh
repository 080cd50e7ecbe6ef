#!/bin/bash
# Build a variant of libmodet_hip.so: ONE source recompiled with extra flags, linked with the product build's other objects.
#   bash tools/build_variant.sh <name> <source.hip> "<extra flags>"   ->  build/variants/libmodet_hip_<name>.so
# (build/ is git-ignored but travels with gpurun snapshots; select it with MODET_HIP_LIB=build/variants/libmodet_hip_<name>.so)
set -e
R=$(cd "$(dirname "$0")/.." && pwd)
name=$1; src=$2; flags=$3
mkdir -p $R/build/variants/obj_$name
base=$(basename $src .hip)
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -munsafe-fp-atomics -fno-gpu-rdc -fno-slp-vectorize $flags -c $R/smilecode_amd/csrc/$base.hip -o $R/build/variants/obj_$name/$base.o
objs=""
for o in $R/smilecode_amd/lib/obj/*.o; do
  b=$(basename $o)
  if [ "$b" == "$base.o" ]; then objs="$objs $R/build/variants/obj_$name/$base.o"; else objs="$objs $o"; fi
done
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC $objs -o $R/build/variants/libmodet_hip_$name.so
echo built $R/build/variants/libmodet_hip_$name.so
