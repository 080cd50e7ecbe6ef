#!/bin/bash
# Run ON THE GPU BOX: per-kernel durations and HBM traffic of the operator boundary (modet_fw / modet_bw) at the level-1
# shape (160x192x160, 1 head, head_dim 6).  Kernel trace -> tools/trace_stats.py; FETCH_SIZE / WRITE_SIZE in separate
# --pmc passes -> tools/pmc_traffic.py (gfx950 correction inside).
set -u
R=${GRAFT_REPO_ROOT:-/root/repo}
TAG=${1:-r02f}
OUT=$R/gpurun_out/profiles_new
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
CMD="python $R/tools/bench_operator.py --level 1 --iters 10"
timeout 200 rocprofv3 --kernel-trace --output-format csv -d /tmp/op_t -o tr -- $CMD > /dev/null 2>&1
python $R/tools/trace_stats.py $(find /tmp/op_t -name "*kernel_trace.csv" | head -1) --csv $OUT/${TAG}_kernel_stats_operator_160x192x160.csv --top 12
timeout 200 rocprofv3 --pmc FETCH_SIZE --output-format csv -d /tmp/op_f -o t -- $CMD > /dev/null 2>&1
timeout 200 rocprofv3 --pmc WRITE_SIZE --output-format csv -d /tmp/op_w -o t -- $CMD > /dev/null 2>&1
python $R/tools/pmc_traffic.py $(find /tmp/op_f -name "*counter_collection.csv" | head -1) $(find /tmp/op_w -name "*counter_collection.csv" | head -1) \
  $OUT/${TAG}_pmc_traffic_operator.json $OUT/${TAG}_pmc_traffic_operator.csv
cat $OUT/${TAG}_pmc_traffic_operator.csv | head -12
