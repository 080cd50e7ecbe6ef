#!/bin/bash
# Run ON THE GPU BOX: rocprofv3 kernel trace of a command, then per-(kernel, grid) averages (tools/trace_by_grid.py).
#   bash tools/prof_by_grid.sh <name-filter> <command...>
R=${GRAFT_REPO_ROOT:-/root/repo}
flt=$1; shift
cd /tmp && export TMPDIR=/tmp
d=$(mktemp -d /tmp/pbg.XXXX)
timeout 600 rocprofv3 --kernel-trace --output-format csv -d $d -o tr -- "$@" > $d/out.txt 2>&1 < /dev/null
python $R/tools/trace_by_grid.py $(find $d -name "*kernel_trace.csv" | head -1) "$flt"
