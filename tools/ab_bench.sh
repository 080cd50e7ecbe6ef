#!/bin/bash
# A/B of the whole train step on ONE box: bench.py alternately with the product library and each variant library.
#   bash tools/ab_bench.sh <rounds> <variant name> [<variant name> ...]      (variants: build/variants/libmodet_hip_<name>.so)
# Prints "<name> <pairs/s> <ms/step>" per run.
R=$(cd "$(dirname "$0")/.." && pwd)
rounds=$1; shift
one() {
  python $R/bench.py --steps 40 --warmup 8 --no-cpu-baseline --no-extra $BENCH_ARGS 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.readlines()[-1]); print('$1', round(d['value'],2), round(d['ms_per_step'],4))"
}
for r in $(seq $rounds); do
  one base
  for v in "$@"; do MODET_HIP_LIB=$R/build/variants/libmodet_hip_$v.so one $v; done
done
