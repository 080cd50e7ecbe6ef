#!/bin/bash
# kernel trace of the eager train step -> per-kernel stats CSV (first part of tools/refresh_profiles.sh)
#   bash tools/trace_only.sh <tag> [extra bench args]
R=${GRAFT_REPO_ROOT:-/root/repo}
TAG=${1:-x}; shift
OUT=$R/gpurun_out/profiles_new
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/prof_t
timeout 300 rocprofv3 --kernel-trace --output-format csv -d /tmp/prof_t -o tr -- python $R/bench.py --steps 8 --warmup 3 --no-cpu-baseline --no-extra --graph off "$@" > /dev/null 2>&1
f=$(find /tmp/prof_t -name "*kernel_trace.csv" | head -1)
python $R/tools/trace_stats.py $f --csv $OUT/${TAG}_kernel_stats.csv --top 40
