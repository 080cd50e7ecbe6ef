#!/bin/bash
# interleaved A/B: exact-fp32 MFMA convs (default) vs the bf16x3 emulation (MODET_CONV_SPLIT=1), separate processes
N=${1:-8}
a=""; b=""
for i in $(seq 1 $N); do
  v=$(python /root/repo/bench.py --steps 30 --warmup 10 --no-cpu-baseline --no-extra 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('%.2f' % d['ms_per_step'])")
  a="$a $v"
  v=$(MODET_CONV_SPLIT=1 python /root/repo/bench.py --steps 30 --warmup 10 --no-cpu-baseline --no-extra 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('%.2f' % d['ms_per_step'])")
  b="$b $v"
done
echo "exact fp32 MFMA :$a"
echo "bf16x3 split    :$b"
