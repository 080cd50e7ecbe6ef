#!/bin/bash
# Run ON THE GPU BOX: build a tuning copy of the library (phase counters compiled in) next to the product one and print
# the per-phase cycle shares of conv_x3_kernel for the level-1 layers.
set -u
R=${GRAFT_REPO_ROOT:-/root/repo}
mkdir -p /tmp/x3t/obj
cd $R/smilecode_amd/csrc
for f in *.hip; do
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -munsafe-fp-atomics -fno-gpu-rdc -DMODET_TUNING -c $f -o /tmp/x3t/obj/${f%.hip}.o &
done
wait
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC /tmp/x3t/obj/*.o -o /tmp/x3t/libmodet_tuning.so
cd $R
for cfg in "fwd 8 8" "fwd 4 8" "wgrad 8 8" "wgrad 4 8"; do
  MODET_HIP_LIB=/tmp/x3t/libmodet_tuning.so python tools/exp_x3_phases.py $cfg
done
