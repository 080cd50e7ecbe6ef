#!/usr/bin/env python
"""Per-kernel stats (calls, total/avg/min/max duration, share) from a rocprofv3 rocpd SQLite database
(rocprofv3 --kernel-trace ... writes <dir>/*_results.db).  Equivalent of `--stats`' kernel_stats.csv.

    python tools/rocpd_stats.py gpurun_out/prof/trace_results.db [--csv out.csv] [--last-steps K --steps-total N]
"""
import argparse
import re
import sqlite3
import sys


def short(name):
    name = re.sub(r"\(anonymous namespace\)::", "", name)
    name = re.sub(r"^void ", "", name)
    return name if len(name) < 110 else name[:107] + "..."


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("db")
    ap.add_argument("--csv", default="")
    ap.add_argument("--top", type=int, default=60)
    args = ap.parse_args()
    db = sqlite3.connect(args.db)
    rows = db.execute("select name, duration, grid_x, grid_y, grid_z, workgroup_x, lds_size, vgpr_count from kernels").fetchall()
    agg = {}
    for name, dur, gx, gy, gz, wx, lds, vg in rows:
        d = agg.setdefault(name, [0, 0, 1 << 62, 0, lds, vg])
        d[0] += 1; d[1] += dur; d[2] = min(d[2], dur); d[3] = max(d[3], dur)
    tot = sum(d[1] for d in agg.values())
    lines = ["Name,Calls,TotalDurationNs,AverageNs,MinNs,MaxNs,Percentage,LDS,VGPR"]
    for name, d in sorted(agg.items(), key=lambda kv: -kv[1][1]):
        lines.append('"%s",%d,%d,%.1f,%d,%d,%.3f,%s,%s' % (short(name), d[0], d[1], d[1] / d[0], d[2], d[3], 100.0 * d[1] / tot, d[4], d[5]))
    if args.csv:
        open(args.csv, "w").write("\n".join(lines) + "\n")
    print(f"# {len(rows)} dispatches, {tot / 1e6:.3f} ms total kernel time")
    for l in lines[:args.top + 1]:
        print(l)


if __name__ == "__main__":
    main()
