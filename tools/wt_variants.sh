#!/bin/bash
# ON THE GPU BOX: kernel times of the level-<CALL> tile-path call for library variants (build/variants/libmodet_hip_<name>.so; "base" = product)
R=${GRAFT_REPO_ROOT:-/root/repo}
for v in "$@"; do
  echo "== $v (call ${CALL:-0})"
  lib=$R/build/variants/libmodet_hip_$v.so; [ "$v" == base ] && lib=$R/smilecode_amd/lib/libmodet_hip.so
  MODET_HIP_LIB=$lib CALL=${CALL:-0} timeout 200 bash $R/tools/prof_kernels.sh wt_$v python $R/tools/warp_real.py tiles 20 2>&1 | grep -i "count_k\|scan\|fill\|accum\|border"
done
