#!/bin/bash
# Run ON THE GPU BOX: A/B of conv3d_x3.hip compile-time variants (-DX3_VARIANT=n) through tools/bench_conv.py.
#   bash tools/x3_variants.sh 0 1 2 3
set -u
R=${GRAFT_REPO_ROOT:-/root/repo}
mkdir -p /tmp/x3v
cd $R/smilecode_amd/csrc
FL="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -munsafe-fp-atomics -fno-gpu-rdc"
OBJS=""
for f in *.hip; do
  [ $f = conv3d_x3.hip ] && continue
  /opt/rocm/bin/hipcc $FL -c $f -o /tmp/x3v/${f%.hip}.o 2>/dev/null &
  OBJS="$OBJS /tmp/x3v/${f%.hip}.o"
done
for v in "$@"; do
  /opt/rocm/bin/hipcc $FL -DX3_VARIANT=$v -c conv3d_x3.hip -o /tmp/x3v/x3_$v.o 2>/dev/null &
done
wait
for v in "$@"; do /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC $OBJS /tmp/x3v/x3_$v.o -o /tmp/x3v/lib_$v.so & done
wait
cd $R
for v in "$@"; do
  echo "== X3_VARIANT=$v"
  MODET_HIP_LIB=/tmp/x3v/lib_$v.so python tools/bench_conv.py 2>&1 | grep -v amdgpu.ids | head -${LINES_PER:-2}
done
