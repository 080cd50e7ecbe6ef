#!/bin/bash
# Run ON THE GPU BOX: per-kernel durations of any command (rocprofv3 kernel trace, csv), summed per kernel name.
#   bash tools/prof_kernels.sh <out-name> <command...>      -> gpurun_out/<out-name>_kernels.txt
R=${GRAFT_REPO_ROOT:-/root/repo}
name=$1; shift
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/prof_k
timeout 400 rocprofv3 --kernel-trace --output-format csv -d /tmp/prof_k -o tr -- "$@" > /tmp/prof_k.out 2>/tmp/prof_k.err < /dev/null
tail -6 /tmp/prof_k.out
f=$(find /tmp/prof_k -name '*kernel_trace.csv' | head -1)
if [ -z "$f" ]; then echo "no kernel trace"; tail -5 /tmp/prof_k.err; exit 1; fi
python3 - "$f" > $R/gpurun_out/${name}_kernels.txt <<'PY'
import csv, sys, collections, re
agg = collections.defaultdict(lambda: [0, 0.0])
for r in csv.DictReader(open(sys.argv[1])):
    n = re.sub(r"\(anonymous namespace\)::", "", r["Kernel_Name"])
    a = agg[n]; a[0] += 1; a[1] += (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3
for n, (c, t) in sorted(agg.items(), key=lambda kv: -kv[1][1])[:40]:
    print("%6d calls %10.1f us total %9.2f us avg  %s" % (c, t, t / c, n[:110]))
PY
head -12 $R/gpurun_out/${name}_kernels.txt
