"""The attention backward ALONE under the condition that breaks the 2-rank test (another process replaying train steps on the
same GPU + a few extra high-/normal-priority streams in both): modet_na_bwd on fixed inputs, N times, every output compared
BIT FOR BIT with the first call's (the kernel has no atomics).  Variants: --between runs a memory-bound filler kernel between
the calls (the op's neighbours in the real step), --fresh allocates new output buffers for every call (as autograd does).

    python tools/exp_na_iso.py [--iters N] [--noise 1] [--streams 4] [--between] [--fresh] [--shape 32,48,32] [--heads 1]
"""
import argparse
import os
import subprocess
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from smilecode_amd import _lib, ops                           # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--iters", type=int, default=20000)
ap.add_argument("--noise", type=int, default=1)
ap.add_argument("--streams", type=int, default=4)
ap.add_argument("--between", action="store_true")
ap.add_argument("--fresh", action="store_true")
ap.add_argument("--shape", default="32,48,32")
ap.add_argument("--heads", type=int, default=1)
ap.add_argument("--analyze", type=int, default=0, help="describe the wrong elements of the first N bad calls (syncs every call)")
args = ap.parse_args()
dev = torch.device("cuda")
D, H, W = (int(s) for s in args.shape.split(","))
heads = args.heads
g = torch.Generator().manual_seed(3)
q = torch.randn(1, D, H, W, heads * 6, generator=g).to(dev)
k = torch.randn(1, D, H, W, heads * 6, generator=g).to(dev)
rpb = (torch.randn(heads, 3, 3, 3, generator=g) * 0.3).to(dev)
dout = (torch.randn(1, D, H, W, heads * 3, generator=g) * 1e-3).to(dev)
L = _lib.load()
P = ops._p
S = ops._stream
out = torch.empty(1, D, H, W, heads * 3, device=dev)
lse = torch.empty(1, D, H, W, heads, device=dev)
_lib.check(L.modet_na_fwd(P(q), P(k), P(rpb), P(out), P(lse), 1, D, H, W, heads, 6, 1.0, S()), "fwd")
nb = L.modet_na_bwd_ws_bytes(1, D, H, W, heads)


def bwd(dq, dk, drpb, ws):
    _lib.check(L.modet_na_bwd(P(q), P(k), P(rpb), P(out), P(lse), P(dout), P(dq), P(dk), P(drpb), P(ws), nb, 1, D, H, W, heads, 6,
                              1.0, S()), "bwd")


extra = [torch.cuda.Stream(priority=-1) for _ in range(args.streams)] + [torch.cuda.Stream() for _ in range(args.streams)]
tick = [torch.zeros(256, device=dev) for _ in extra]


def poke():
    for st, t in zip(extra, tick):
        with torch.cuda.stream(st):
            t.add_(1.0)


dq0, dk0, dr0 = torch.empty_like(q), torch.empty_like(k), torch.empty_like(rpb)
ws0 = torch.empty(nb // 4 + 1, device=dev)
bwd(dq0, dk0, dr0, ws0)
torch.cuda.synchronize()
noise = [subprocess.Popen([sys.executable, os.path.join(os.path.dirname(os.path.abspath(__file__)), "race_hunt.py"), "--as-noise",
                           str(14 + args.iters * 0.0004), "--streams", str(args.streams)], stdout=subprocess.DEVNULL,
                          stderr=subprocess.DEVNULL) for _ in range(args.noise)]
if noise:
    time.sleep(12)
filler = torch.randn(4_000_000, device=dev)
dq, dk, dr, ws = torch.empty_like(q), torch.empty_like(k), torch.empty_like(rpb), torch.empty(nb // 4 + 1, device=dev)
bad = []
t0 = time.time()
for i in range(args.iters):
    if args.fresh:
        dq, dk, dr, ws = torch.empty_like(q), torch.empty_like(k), torch.empty_like(rpb), torch.empty(nb // 4 + 1, device=dev)
    if args.between:
        filler.mul_(1.0000001)
    bwd(dq, dk, dr, ws)
    if args.between:
        filler.add_(1e-9)
    bad.append(torch.stack([(dq != dq0).sum(), (dk != dk0).sum(), (dr != dr0).sum()]))
    if args.analyze and int(bad[-1].sum()) > 0:
        args.analyze -= 1
        idx = torch.nonzero(dq != dq0).cpu()
        vox = sorted(set((int(r[1]), int(r[2]), int(r[3])) for r in idx))
        print("call %d: %d wrong d_q elements in %d voxels; (z,y,x) -> tile thread (tz,ty,tx) / lane:" % (i, idx.shape[0], len(vox)))
        for z, y, x in vox[:40]:
            tz, ty, tx = z % 4, y % 4, x % 16
            a, b = dq0[0, z, y, x].cpu(), dq[0, z, y, x].cpu()
            print("   (%2d,%2d,%2d) tile (%d,%d,%d) thread (%d,%d,%2d) wave %d lane %2d  ref %s  got %s" % (
                z, y, x, z // 4, y // 4, x // 16, tz, ty, tx, tz, ty * 16 + tx, ["% .3e" % v for v in a.tolist()], ["% .3e" % v for v in b.tolist()]))
        ir = torch.nonzero(dr != dr0).cpu()
        print("   wrong d_rpb taps:", [(tuple(r.tolist()), float(dr0[tuple(r.tolist())]), float(dr[tuple(r.tolist())])) for r in ir[:8]])
    if i % 4 == 0:
        poke()
    if i % 500 == 499:
        torch.cuda.synchronize()
torch.cuda.synchronize()
dt = time.time() - t0
B = torch.stack(bad).cpu()
hits = torch.nonzero(B.sum(1)).flatten().tolist()
alive = sum(p.poll() is None for p in noise)
print("na_bwd alone (%dx%dx%d, %d head%s%s%s), %d noise processes (%d alive at the end), %d+%d streams: %d calls in %.1f s; calls whose outputs "
      "are not bit-identical to the first: %d  (d_q %d, d_k %d, d_rpb %d)  first %s" % (
          D, H, W, heads, "s" if heads > 1 else "", ", filler kernels between" if args.between else "", ", fresh outputs" if args.fresh else "",
          len(noise), alive, args.streams, args.streams, args.iters, dt, len(hits), int((B[:, 0] > 0).sum()), int((B[:, 1] > 0).sum()),
          int((B[:, 2] > 0).sum()), [(i, B[i].tolist()) for i in hits[:6]]))
for p in noise:
    p.kill()
