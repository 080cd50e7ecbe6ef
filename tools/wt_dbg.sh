#!/bin/bash
# ON THE GPU BOX: kernel times of the level-1 tile-path call with parts left out (tuning variant build, WT_DBG bit mask; results are wrong)
R=${GRAFT_REPO_ROOT:-/root/repo}
for d in "$@"; do
  echo "== WT_DBG=$d"
  timeout 120 env WT_DBG=$d MODET_HIP_LIB=$R/build/variants/libmodet_hip_tune.so CALL=${CALL:-0} bash $R/tools/prof_kernels.sh wt_dbg$d python $R/tools/warp_real.py tiles 20 2>&1 | grep -i "fill\|accum"
done
