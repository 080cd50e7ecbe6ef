#!/usr/bin/env python
"""Per-kernel HBM traffic from two rocprofv3 --pmc passes (FETCH_SIZE, WRITE_SIZE) -> profiles/pmc_traffic.json.

    rocprofv3 --pmc FETCH_SIZE --output-format csv -d DIR_F -o t -- python bench.py --steps 2 --warmup 2 --no-cpu-baseline
    rocprofv3 --pmc WRITE_SIZE --output-format csv -d DIR_W -o t -- python bench.py --steps 2 --warmup 2 --no-cpu-baseline
    python tools/pmc_traffic.py FETCH_counter_collection.csv WRITE_counter_collection.csv out.json [out.csv]

Counters are KB; per MI355X_MICROARCH.md (HBM / rocprofv3 section) gfx950's FETCH_SIZE reports half of wide coalesced
reads, so hbm_bytes = 2 * FETCH_SIZE * 1024 + WRITE_SIZE * 1024.  Values are means over all launches of a kernel symbol
(template arguments stripped), i.e. per launch like bench.py's roofline.achieved."""
import csv
import hashlib
import json
import os
import re
import sys
from collections import defaultdict


def csrc_sha16():
    """fingerprint of the kernel sources this profile was taken on (bench.py recomputes it: a profile older than the kernels
    is reported as stale instead of being quoted as roofline.traffic; the GPU box has no .git, so a commit id is not available there)"""
    root = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "smilecode_amd", "csrc")
    h = hashlib.sha256()
    for fn in sorted(os.listdir(root)):
        if fn.endswith((".hip", ".h")):
            h.update(fn.encode())
            h.update(open(os.path.join(root, fn), "rb").read())
    return h.hexdigest()[:16]


def family(name):
    name = re.sub(r"\(anonymous namespace\)::", "", name)
    name = re.sub(r"^void ", "", name)
    name = re.sub(r"\(.*$", "", name)
    return re.sub(r"<.*$", "", name)


def load(path, counter):
    acc = defaultdict(list)
    with open(path) as f:
        for r in csv.DictReader(f):
            if r["Counter_Name"] == counter:
                acc[family(r["Kernel_Name"])].append(float(r["Counter_Value"]) * 1024.0)
    return acc


fetch, write = load(sys.argv[1], "FETCH_SIZE"), load(sys.argv[2], "WRITE_SIZE")
# steps profiled = launches of the optimizer kernel (one per train step); a forward-only run has none: per-step fields stay null
steps = max(len(fetch.get("adam_kernel", [])), len(write.get("adam_kernel", [])))
fams = {}
for k in sorted(set(fetch) | set(write), key=lambda k: -(2 * sum(fetch.get(k, [0])) + sum(write.get(k, [0])))):
    f, w = fetch.get(k, []), write.get(k, [])
    n = max(len(f), len(w))
    fm, wm = (sum(f) / len(f) if f else 0.0), (sum(w) / len(w) if w else 0.0)
    fams[k] = {"launches_profiled": n, "launches_per_step": (n / steps if steps else None), "fetch_bytes_raw": fm, "write_bytes": wm,
               "hbm_bytes_per_launch_corrected": 2 * fm + wm}
out = {"note": "rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE in two separate passes over `bench.py --steps 2 --warmup 2` (train "
               "step, 160x192x160, B=1); counters are KB; hbm_bytes = 2*FETCH_SIZE*1024 + WRITE_SIZE*1024 per "
               "MI355X_MICROARCH.md (gfx950 FETCH_SIZE reports 1/2 of wide coalesced reads); averaged over all launches of "
               "the kernel symbol in a step",
       "csrc_sha16": csrc_sha16(),
       "steps_profiled": steps,
       "families": fams}
for k in fams:                                   # bench.py reads the dominant family's per-launch bytes from the top level
    out[k] = fams[k]["hbm_bytes_per_launch_corrected"]
json.dump(out, open(sys.argv[3], "w"), indent=1)
if len(sys.argv) > 4:
    with open(sys.argv[4], "w") as f:
        f.write("kernel_family,launches_profiled,FETCH_SIZE_bytes_per_launch_raw,WRITE_SIZE_bytes_per_launch,"
                "hbm_bytes_per_launch_corrected(2*fetch+write)\n")
        for k, v in fams.items():
            f.write("%s,%d,%.0f,%.0f,%.0f\n" % (k, v["launches_profiled"], v["fetch_bytes_raw"], v["write_bytes"],
                                                v["hbm_bytes_per_launch_corrected"]))
print(json.dumps({k: round(v["hbm_bytes_per_launch_corrected"] / 1e6, 1) for k, v in list(fams.items())[:12]}))
