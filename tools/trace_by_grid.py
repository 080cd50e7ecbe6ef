#!/usr/bin/env python
"""Average duration per (kernel, grid size) from a rocprofv3 kernel trace csv, in first-dispatch order: separates the launches of one
kernel symbol that belong to different layers.   python tools/trace_by_grid.py <kernel_trace.csv> [name-filter]"""
import csv
import re
import sys

agg, order = {}, []
flt = sys.argv[2] if len(sys.argv) > 2 else ""
for r in csv.DictReader(open(sys.argv[1])):
    n = re.sub(r"\(anonymous namespace\)::", "", r["Kernel_Name"])
    n = re.sub(r"^void ", "", n)
    n = re.sub(r"\(.*$", "", n)
    if flt and flt not in n:
        continue
    k = (n, r.get("Grid_Size", r.get("Grid_Size_X", "")), r.get("LDS_Block_Size", ""), r.get("VGPR_Count", ""))
    if k not in agg:
        agg[k] = [0, 0.0, 1e30]
        order.append(k)
    d = (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3
    agg[k][0] += 1; agg[k][1] += d; agg[k][2] = min(agg[k][2], d)
for k in order:
    c, t, mn = agg[k]
    print("%5d calls  avg %9.2f us  min %9.2f us  grid %-9s lds %-6s vgpr %-4s %s" % (c, t / c, mn, k[1], k[2], k[3], k[0][:90]))
