#!/usr/bin/env python
"""Aggregate a rocprofv3 --pmc counter_collection.csv per kernel: mean counter value per dispatch.
    python tools/pmc_summary.py gpurun_out/pmc_x/t_counter_collection.csv [kernel-substring]"""
import csv
import sys
from collections import defaultdict

path = sys.argv[1]
flt = sys.argv[2] if len(sys.argv) > 2 else ""
acc = defaultdict(lambda: defaultdict(list))
with open(path) as f:
    for r in csv.DictReader(f):
        k = r["Kernel_Name"]
        if flt and flt not in k:
            continue
        acc[k[:70]][r["Counter_Name"]].append(float(r["Counter_Value"]))
for k, d in acc.items():
    print(k)
    for c, v in sorted(d.items()):
        print(f"   {c:32s} n={len(v):3d} mean={sum(v) / len(v):16.1f}")
