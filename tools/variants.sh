#!/bin/bash
# Run ON THE GPU BOX: A/B of compile-time variants of ONE source file (-D<MACRO>=n), each linked into its own library.
#   bash tools/variants.sh losses.hip NCC_VARIANT "python tools/bench_ncc.py" 0 1 2 4
set -u
R=${GRAFT_REPO_ROOT:-/root/repo}
SRC=$1; MACRO=$2; CMD=$3; shift 3
D=/tmp/variants; mkdir -p $D
cd $R/smilecode_amd/csrc
FL="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -munsafe-fp-atomics -fno-gpu-rdc"
OBJS=""
for f in *.hip; do
  [ $f = $SRC ] && continue
  /opt/rocm/bin/hipcc $FL -c $f -o $D/${f%.hip}.o 2>/dev/null < /dev/null &
  OBJS="$OBJS $D/${f%.hip}.o"
done
for v in "$@"; do /opt/rocm/bin/hipcc $FL -D$MACRO=$v -c $SRC -o $D/v_$v.o 2>/dev/null < /dev/null & done
wait
for v in "$@"; do /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC $OBJS $D/v_$v.o -o $D/lib_$v.so < /dev/null & done
wait
cd $R
for v in "$@"; do
  echo "== $MACRO=$v"
  MODET_HIP_LIB=$D/lib_$v.so timeout 300 $CMD 2>&1 < /dev/null | grep -v amdgpu.ids | head -${LINES_PER:-4}
done
