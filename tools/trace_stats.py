#!/usr/bin/env python
"""Per-kernel stats from a rocprofv3 --kernel-trace --output-format csv trace (kernel_trace.csv):
the same table `rocprofv3 --stats` prints (calls, total/avg/min/max ns, share).
    python tools/trace_stats.py gpurun_out/prof/trace_kernel_trace.csv [--csv out.csv] [--top N] [--skip-first K]"""
import argparse
import csv
import re


def short(name):
    name = re.sub(r"\(anonymous namespace\)::", "", name)
    name = re.sub(r"^void ", "", name)
    name = re.sub(r"\(.*$", "", name)
    return name if len(name) < 100 else name[:97] + "..."


ap = argparse.ArgumentParser()
ap.add_argument("trace")
ap.add_argument("--csv", default="")
ap.add_argument("--top", type=int, default=50)
args = ap.parse_args()
agg = {}
n = 0
with open(args.trace) as f:
    for r in csv.DictReader(f):
        dur = int(r["End_Timestamp"]) - int(r["Start_Timestamp"])
        k = short(r["Kernel_Name"])
        d = agg.setdefault(k, [0, 0, 1 << 62, 0, r.get("LDS_Block_Size", ""), r.get("VGPR_Count", ""), r.get("Accum_VGPR_Count", "")])
        d[0] += 1; d[1] += dur; d[2] = min(d[2], dur); d[3] = max(d[3], dur)
        n += 1
tot = sum(d[1] for d in agg.values())
lines = ["Name,Calls,TotalDurationNs,AverageNs,MinNs,MaxNs,Percentage,LDS,VGPR,AGPR"]
for k, d in sorted(agg.items(), key=lambda kv: -kv[1][1]):
    lines.append('"%s",%d,%d,%.1f,%d,%d,%.3f,%s,%s,%s' % (k, d[0], d[1], d[1] / d[0], d[2], d[3], 100.0 * d[1] / tot, d[4], d[5], d[6]))
if args.csv:
    open(args.csv, "w").write("\n".join(lines) + "\n")
print(f"# {n} dispatches, {tot / 1e6:.3f} ms total kernel time")
for l in lines[:args.top + 1]:
    print(l)
