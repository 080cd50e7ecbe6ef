#!/bin/bash
# Run ON THE GPU BOX: every launch of the kernels matching a pattern in ONE eager train step, in launch order: duration, grid, LDS.
#   bash tools/prof_calls.sh <out-name> <kernel-substring>      -> gpurun_out/<out-name>_calls.txt
R=${GRAFT_REPO_ROOT:-/root/repo}
name=$1; pat=$2
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/prof_c
timeout 300 rocprofv3 --kernel-trace --output-format csv -d /tmp/prof_c -o tr -- python $R/bench.py --steps 2 --warmup 2 --no-cpu-baseline --no-extra --graph off > /dev/null 2>/tmp/prof_c.err < /dev/null
f=$(find /tmp/prof_c -name '*kernel_trace.csv' | head -1)
if [ -z "$f" ]; then echo "no kernel trace"; tail -5 /tmp/prof_c.err; exit 1; fi
python3 - "$f" "$pat" > $R/gpurun_out/${name}_calls.txt <<'PY'
import csv, sys
rows = [r for r in csv.DictReader(open(sys.argv[1]))]
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
sel = [r for r in rows if sys.argv[2] in r["Kernel_Name"]]
n = len(sel)
per = n // 4 if n >= 4 else n                      # 2 warm-up + 2 timed steps
for r in sel[-per:]:
    print("%9.1f us  grid %-14s wg %-6s lds %-6s %s" % ((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3,
          "x".join([r.get("Grid_Size_X", "?"), r.get("Grid_Size_Y", "?"), r.get("Grid_Size_Z", "?")]), r.get("Workgroup_Size_X", "?"),
          r.get("LDS_Block_Size", "?"), r["Kernel_Name"][:70]))
PY
cat $R/gpurun_out/${name}_calls.txt
