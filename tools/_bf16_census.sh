cd /tmp && export TMPDIR=/tmp
timeout 300 rocprofv3 --kernel-trace --output-format csv -d /tmp/pb -o tr -- python /root/repo/bench.py --dtype bf16 --steps 6 --warmup 3 --no-cpu-baseline --no-extra --graph off > /dev/null 2>&1
f=$(find /tmp/pb -name "*kernel_trace.csv" | head -1)
python /root/repo/tools/step_gaps.py $f --census | tail -45
python /root/repo/tools/trace_stats.py $f --top 25 | cut -c1-150
