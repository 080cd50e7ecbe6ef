#!/bin/bash
# Run ON THE GPU BOX (gpurun): regenerates the artefacts under gpurun_out/profiles_new/ that get copied into profiles/.
#   kernel trace -> per-kernel stats CSV (rocprofv3 --kernel-trace; --stats hung in post-processing on this image, so
#   the statistics are computed from the trace by tools/trace_stats.py), two --pmc passes -> pmc_traffic.json,
#   one default bench run with the per-op breakdown.
set -u
R=${GRAFT_REPO_ROOT:-/root/repo}
TAG=${1:-r01e}
OUT=$R/gpurun_out/profiles_new
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
timeout 200 rocprofv3 --kernel-trace --output-format csv -d /tmp/prof_t -o tr -- python $R/bench.py --steps 8 --warmup 3 --no-cpu-baseline --no-extra --graph off > /dev/null 2>&1
f=$(find /tmp/prof_t -name "*kernel_trace.csv" | head -1)
python $R/tools/trace_stats.py $f --csv $OUT/${TAG}_kernel_stats_train_160x192x160.csv --top 12
timeout 200 rocprofv3 --pmc FETCH_SIZE --output-format csv -d /tmp/prof_f -o t -- python $R/bench.py --steps 2 --warmup 2 --no-cpu-baseline --no-extra --graph off > /dev/null 2>&1
timeout 200 rocprofv3 --pmc WRITE_SIZE --output-format csv -d /tmp/prof_w -o t -- python $R/bench.py --steps 2 --warmup 2 --no-cpu-baseline --no-extra --graph off > /dev/null 2>&1
ff=$(find /tmp/prof_f -name "*counter_collection.csv" | head -1)
fw=$(find /tmp/prof_w -name "*counter_collection.csv" | head -1)
python $R/tools/pmc_traffic.py $ff $fw $OUT/pmc_traffic.json $OUT/${TAG}_pmc_traffic_by_kernel.csv
cd $R
cp $OUT/pmc_traffic.json $R/profiles/pmc_traffic.json      # bench.py reads roofline.traffic from here
timeout 400 python bench.py --breakdown $OUT/${TAG}_breakdown_train_160x192x160.json > $OUT/${TAG}_bench.json 2> $OUT/${TAG}_bench.err
cat $OUT/${TAG}_bench.json
