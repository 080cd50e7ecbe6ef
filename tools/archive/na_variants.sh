#!/bin/bash
# Run ON THE GPU BOX: tuning build; old tile kernel vs z-marching kernel, and the workgroup target of the latter
set -u
R=${GRAFT_REPO_ROOT:-/root/repo}
mkdir -p /tmp/nat
cd $R/smilecode_amd/csrc
FL="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -munsafe-fp-atomics -fno-gpu-rdc -DMODET_TUNING"
for f in *.hip; do /opt/rocm/bin/hipcc $FL -c $f -o /tmp/nat/${f%.hip}.o 2>/dev/null & done
wait
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC /tmp/nat/*.o -o /tmp/nat/lib.so
cd $R
echo "== old tile kernel"; MODET_NA_MARCH=0 MODET_HIP_LIB=/tmp/nat/lib.so python tools/bench_na.py 2>&1 | grep -v amdgpu
for w in 1024 2000 3000; do
  echo "== march, MODET_NA_WGS=$w"; MODET_NA_WGS=$w MODET_HIP_LIB=/tmp/nat/lib.so python tools/bench_na.py 2>&1 | grep -v amdgpu | head -1
done
