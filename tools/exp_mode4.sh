#!/bin/bash
# Run ON THE GPU BOX: N un-profiled processes of the default bench with its own HIP-event per-op breakdown; which op
# families differ between the fastest and the slowest process?
N=${1:-6}
for i in $(seq 1 $N); do
  python /root/repo/bench.py --steps 30 --warmup 10 --no-cpu-baseline --no-extra --breakdown /tmp/bd_$i.json > /tmp/b_$i.json 2>/dev/null
done
python - <<PY
import json
N=$N
ms=[json.load(open("/tmp/b_%d.json"%i))["ms_per_step"] for i in range(1,N+1)]
bd=[json.load(open("/tmp/bd_%d.json"%i))["families"] for i in range(1,N+1)]
print("ms/step:", ["%.2f"%m for m in ms])
f=min(range(N), key=lambda i: ms[i]); s=max(range(N), key=lambda i: ms[i])
keys=sorted(bd[f], key=lambda k: -(bd[s].get(k,{"ms":0})["ms"]-bd[f][k]["ms"]))
print("%-28s %8s %8s %8s" % ("family (ms/step)", "fast", "slow", "ratio"))
for k in keys:
    a,b=bd[f][k]["ms"], bd[s].get(k,{"ms":0})["ms"]
    print("%-28s %8.3f %8.3f %8.3f" % (k, a, b, b/a if a else 0))
print("sum", sum(v["ms"] for v in bd[f].values()), sum(v["ms"] for v in bd[s].values()))
PY
