#!/usr/bin/env python
"""Per-step timeline occupancy from a rocprofv3 kernel trace: steps are delimited by adam_kernel launches.

    python tools/step_gaps.py kernel_trace.csv

For every step: span (first start .. last end), busy (sum of kernel durations), launches, idle = span - busy, and the
share of launches shorter than 12 us.  Used to tell whether the ~200 small launches of a step cost their duration only
(hipGraph replay: gaps ~0) or their duration plus a dispatch gap (eager)."""
import csv
import sys

rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
cuts = [i for i, r in enumerate(rows) if r["Kernel_Name"].startswith("adam_kernel") or "adam_kernel" in r["Kernel_Name"]]
print("# %d dispatches, %d optimizer steps" % (len(rows), len(cuts)))
for a, b in zip(cuts[:-1], cuts[1:]):
    seg = rows[a + 1:b + 1]
    s0, e1 = int(seg[0]["Start_Timestamp"]), max(int(r["End_Timestamp"]) for r in seg)
    dur = [int(r["End_Timestamp"]) - int(r["Start_Timestamp"]) for r in seg]
    small = [d for d in dur if d < 12000]
    gaps = sorted(max(0, int(seg[i + 1]["Start_Timestamp"]) - int(seg[i]["End_Timestamp"])) for i in range(len(seg) - 1))
    print("step: launches %4d  span %7.3f ms  busy %7.3f ms  idle %6.3f ms  small(<12us) %3d launches %6.3f ms  median gap %5d ns  p90 gap %6d ns"
          % (len(seg), (e1 - s0) / 1e6, sum(dur) / 1e6, (e1 - s0 - sum(dur)) / 1e6, len(small), sum(small) / 1e6,
             gaps[len(gaps) // 2], gaps[int(len(gaps) * 0.9)]))

if len(sys.argv) > 2 and sys.argv[2] == "--census" and len(cuts) >= 2:
    import collections
    import re
    a, b = cuts[-2], cuts[-1]
    seg = rows[a + 1:b + 1]
    cnt, tot = collections.Counter(), collections.Counter()
    for r in seg:
        d = int(r["End_Timestamp"]) - int(r["Start_Timestamp"])
        n = re.sub(r"\(anonymous namespace\)::", "", r["Kernel_Name"])
        n = re.sub(r"^void ", "", n)[:70]
        if d < 12000:
            cnt[n] += 1
            tot[n] += d
    print("# launches shorter than 12 us in the last step: %d, %.3f ms" % (sum(cnt.values()), sum(tot.values()) / 1e6))
    for n, c in cnt.most_common(40):
        print("%4d x %6.1f us  %s" % (c, tot[n] / c / 1e3, n))

if len(sys.argv) > 2 and sys.argv[2] == "--timeline" and len(cuts) >= 2:
    # the last step in launch order: offset from the step's start, duration, gap to the previous launch's end, grid, name
    import re
    a, b = cuts[-2], cuts[-1]
    seg = rows[a + 1:b + 1]
    t0 = int(seg[0]["Start_Timestamp"])
    prev_end = t0
    for r in seg:
        s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
        n = re.sub(r"\(anonymous namespace\)::", "", r["Kernel_Name"])
        n = re.sub(r"^void ", "", n)[:90]
        grid = r.get("Grid_Size_X", r.get("Grid_Size", "?"))
        print("%8.1f us  dur %7.1f  gap %6.1f  grid %9s  %s" % ((s - t0) / 1e3, (e - s) / 1e3, (s - prev_end) / 1e3, grid, n))
        prev_end = max(prev_end, e)
