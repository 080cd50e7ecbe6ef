#!/bin/bash
# interleaved A/B of one environment switch on the default bench, separate processes:  exp_ab_env.sh VAR=VALUE [N]
KV=$1; N=${2:-8}
a=""; b=""
for i in $(seq 1 $N); do
  v=$(python /root/repo/bench.py --steps 30 --warmup 10 --no-cpu-baseline --no-extra 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('%.2f' % d['ms_per_step'])")
  a="$a $v"
  v=$(env $KV python /root/repo/bench.py --steps 30 --warmup 10 --no-cpu-baseline --no-extra 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('%.2f' % d['ms_per_step'])")
  b="$b $v"
done
echo "default :$a"
echo "$KV :$b"
