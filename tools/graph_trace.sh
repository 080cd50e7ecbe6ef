#!/bin/bash
# Run ON THE GPU BOX: kernel trace of a hipGraph-replay bench run, per-step occupancy by tools/step_gaps.py
cd /tmp && export TMPDIR=/tmp
timeout 300 rocprofv3 --kernel-trace --output-format csv -d /tmp/prof_g -o tr -- python /root/repo/bench.py --steps 8 --warmup 3 --no-cpu-baseline --no-extra --graph ${1:-on} > /tmp/bench_g.json 2>/tmp/bench_g.err
cut -c1-260 /tmp/bench_g.json
python /root/repo/tools/step_gaps.py $(find /tmp/prof_g -name "*kernel_trace.csv" | head -1) ${2:-} | tail -${3:-60}
