"""compare two bench.py --breakdown files op by op:  python tools/cmp_breakdown.py a.json b.json [min_ms_diff]"""
import json
import sys
a, b = (json.load(open(f))["ops"] for f in sys.argv[1:3])
thr = float(sys.argv[3]) if len(sys.argv) > 3 else 0.004
tot = [0.0, 0.0]
for k in sorted(set(a) | set(b), key=lambda k: -abs(a.get(k, {"ms": 0})["ms"] - b.get(k, {"ms": 0})["ms"])):
    x, y = a.get(k, {"ms": 0.0})["ms"], b.get(k, {"ms": 0.0})["ms"]
    tot[0] += x
    tot[1] += y
    if abs(x - y) >= thr:
        print(f"{k:44s} {x:8.3f} {y:8.3f}  {y - x:+.3f}")
print(f"{'sum':44s} {tot[0]:8.3f} {tot[1]:8.3f}  {tot[1] - tot[0]:+.3f}")
