#!/bin/bash
# Run ON THE GPU BOX: rocprofv3 kernel trace -> per-kernel stats of the bf16-storage train step (cfg 5 on one GPU) + bench lines.
set -u
R=${GRAFT_REPO_ROOT:-/root/repo}
TAG=${1:-r02e}
OUT=$R/gpurun_out/profiles_new
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
timeout 300 rocprofv3 --kernel-trace --output-format csv -d /tmp/prof_b -o tr -- python $R/bench.py --dtype bf16 --steps 8 --warmup 3 --no-cpu-baseline --no-extra --graph off > /dev/null 2>&1
f=$(find /tmp/prof_b -name "*kernel_trace.csv" | head -1)
python $R/tools/trace_stats.py $f --csv $OUT/${TAG}_kernel_stats_train_bf16_160x192x160.csv --top 14
cd $R
timeout 400 python bench.py --dtype bf16 --no-cpu-baseline --no-extra --breakdown $OUT/${TAG}_breakdown_train_bf16_160x192x160.json > $OUT/${TAG}_bench_bf16.json 2> /dev/null
timeout 400 python bench.py --dtype bf16 --shape 160,192,224 --batch 2 --no-cpu-baseline --no-extra > $OUT/${TAG}_bench_cfg5_bf16_160x192x224_b2.json 2> /dev/null
timeout 400 python bench.py --shape 160,192,224 --batch 2 --no-cpu-baseline --no-extra > $OUT/${TAG}_bench_f32_160x192x224_b2.json 2> /dev/null
timeout 400 python bench.py --workload fwd --no-cpu-baseline > $OUT/${TAG}_bench_fwd.json 2> /dev/null
cut -c1-200 $OUT/${TAG}_bench_bf16.json $OUT/${TAG}_bench_cfg5_bf16_160x192x224_b2.json $OUT/${TAG}_bench_f32_160x192x224_b2.json $OUT/${TAG}_bench_fwd.json
