#!/bin/bash
# Run ON THE GPU BOX: kernel traces of N separate processes of the same train step; per-kernel mean durations side by side
# (which kernels make a "slow" 11.9 ms process slower than a "fast" 10.6 ms one?)
N=${1:-5}
cd /tmp && export TMPDIR=/tmp
for i in $(seq 1 $N); do
  rm -rf /tmp/pm_$i
  timeout 300 rocprofv3 --kernel-trace --output-format csv -d /tmp/pm_$i -o tr -- python /root/repo/bench.py --steps 10 --warmup 5 --no-cpu-baseline --no-extra --graph on > /tmp/pm_$i.json 2>/dev/null
done
python - <<PY
import csv, glob, json, collections
N=$N
per=[]; ms=[]
for i in range(1,N+1):
    ms.append(json.load(open("/tmp/pm_%d.json"%i))["ms_per_step"])
    f=glob.glob("/tmp/pm_%d/**/*kernel_trace.csv"%i, recursive=True)[0]
    rows=list(csv.DictReader(open(f)))
    rows.sort(key=lambda r:int(r["Start_Timestamp"]))
    cuts=[k for k,r in enumerate(rows) if "adam_kernel" in r["Kernel_Name"]]
    seg=rows[cuts[-6]+1:cuts[-1]+1]            # the last 5 steps
    d=collections.defaultdict(float)
    for r in seg: d[r["Kernel_Name"][:70]]+= (int(r["End_Timestamp"])-int(r["Start_Timestamp"]))/5e3
    per.append(d)
print("ms/step per process:", ["%.2f"%m for m in ms])
fast=min(range(N), key=lambda i: ms[i]); slow=max(range(N), key=lambda i: ms[i])
keys=sorted(per[fast], key=lambda k: -(per[slow].get(k,0)-per[fast][k]))
print("kernel (us/step)                                                        fast    slow    diff")
for k in keys[:22]:
    print("%-70s %7.1f %7.1f %7.1f" % (k, per[fast][k], per[slow].get(k,0), per[slow].get(k,0)-per[fast][k]))
print("sum fast %.1f slow %.1f" % (sum(per[fast].values()), sum(per[slow].values())))
PY
