#!/usr/bin/env python
"""Print the LAST dispatch sequence of a rocprofv3 kernel trace, one line per dispatch in launch order:
    python tools/trace_seq.py trace_kernel_trace.csv [last_n] [name_filter]"""
import csv
import re
import sys

rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
n = int(sys.argv[2]) if len(sys.argv) > 2 else 60
flt = sys.argv[3] if len(sys.argv) > 3 else ""
rows = [r for r in rows if flt in r["Kernel_Name"]][-n:]
for r in rows:
    name = re.sub(r"\(anonymous namespace\)::", "", r["Kernel_Name"])
    name = re.sub(r"^void ", "", name)
    name = re.sub(r"\(.*$", "", name)[:60]
    print("%-60s %8.1f us  grid %s wg %s lds %s vgpr %s" % (
        name, (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3, r.get("Grid_Size", r.get("Grid_Size_X", "")),
        r.get("Workgroup_Size", r.get("Workgroup_Size_X", "")), r.get("LDS_Block_Size", ""), r.get("VGPR_Count", "")))
