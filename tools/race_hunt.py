"""Race hunt by repetition: the train step as ONE captured hipGraph, replayed N times while other
processes keep the same GPU busy with their own replays (waves of foreign kernels on the same SIMDs: a latent intra-kernel
race that never fires in an idle single-process run gets its chance, as in the 2-rank test).  Every op the model calls is
TAPPED: its forward output and the gradient arriving at it are copied into persistent buffers inside the graph, and after
every replay all taps are compared with the first replay's.  A bad replay names the FIRST tap (forward order, then
backward order) that left the run-to-run noise floor -- the kernel that produced it is the suspect.

    python tools/race_hunt.py [--shape 32,48,32] [--replays 2000] [--noise 1] [--batch 1] [--bf16] [--thresh 2e-5]
"""
import argparse
import os
import subprocess
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from smilecode_amd import models, ops, synth                 # noqa: E402
from smilecode_amd.engine import Trainer                      # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--shape", default="32,48,32")
ap.add_argument("--replays", type=int, default=2000)
ap.add_argument("--noise", type=int, default=1, help="number of co-running processes")
ap.add_argument("--noise-shape", default="")
ap.add_argument("--batch", type=int, default=1)
ap.add_argument("--bf16", action="store_true")
ap.add_argument("--thresh", type=float, default=2e-5)
ap.add_argument("--as-noise", type=float, default=0.0, help="(internal) run as the noise process for this many seconds")
ap.add_argument("--streams", type=int, default=0, help="every process also opens this many high- and normal-priority streams and keeps "
                "tiny kernels going on them (torch.distributed's gloo path opens a pool of 32 + 32): more hardware queues than the "
                "GPU has slots -> the scheduler time-slices the processes' queues by preempting running waves")
ap.add_argument("--dump", default="", help="stop at the first bad replays (up to 4) and save every tap that left its noise floor "
                "(reference + bad values) to this .pt file")
ap.add_argument("--eager", action="store_true", help="run the step eagerly instead of replaying a graph")
args = ap.parse_args()
shape = tuple(int(s) for s in args.shape.split(","))
dev = torch.device("cuda")

kw = dict(act_dtype=torch.bfloat16) if args.bf16 else {}
model = models.ModeT(shape, head_dim=6, num_heads=[8, 4, 2, 1, 1], scale=1, **kw).to(dev)
models.load_numpy_weights(model, synth.make_weights(24))
mov, fix = (torch.from_numpy(a).to(dev) for a in synth.make_pair(shape, 24, args.batch))
tr = Trainer(model)

extra_streams, tick = [], None
if args.streams:
    extra_streams = [torch.cuda.Stream(priority=-1) for _ in range(args.streams)] + [torch.cuda.Stream() for _ in range(args.streams)]
    tick = [torch.zeros(256, device=dev) for _ in extra_streams]


def poke():
    for st, t in zip(extra_streams, tick):
        with torch.cuda.stream(st):
            t.add_(1.0)


if args.as_noise > 0:
    tr.capture(mov, fix, verify=False)
    t0 = time.time()
    n = 0
    while time.time() - t0 < args.as_noise:
        for _ in range(20):
            tr._graph.replay()
            poke()
        torch.cuda.synchronize()
        n += 20
    print("noise process: %d replays" % n)
    sys.exit(0)

# ---------------------------------------------------------------- taps
TAPPED = ["to_channels_last", "to_ncdhw", "warp", "conv3d", "conv3d_instnorm_lrelu", "conv3d_with_stats", "lazy_instnorm_conv3d",
          "instnorm_lrelu_pool_tee_split", "pool_tee", "proj_ln", "proj_ln_pair", "upsample2", "cwm_tail",
          "neighbourhood_attention", "ncc_loss", "grad3d_loss", "level_attention_bf16", "conv_ins_pair_bf16_pool_split",
          "conv_ins_pair_bf16"]
fw, bw, order = {}, {}, []
counter = [0]


def _flat(o):
    if isinstance(o, torch.Tensor):
        yield o
    elif isinstance(o, (tuple, list)):
        for v in o:
            yield from _flat(v)


def tap(name, fn):
    def wrapped(*a, **k):
        out = fn(*a, **k)
        i = counter[0]
        counter[0] += 1
        for j, t in enumerate(_flat(out)):
            if not t.is_floating_point() or t.dtype != torch.float32:
                continue
            key = "%03d %s.%d %s" % (i, name, j, "x".join(map(str, t.shape)))
            if key not in fw:
                fw[key] = torch.zeros_like(t)
                order.append(key)
            fw[key].copy_(t.detach())
            if t.requires_grad:
                if key not in bw:
                    bw[key] = torch.zeros_like(t)
                t.register_hook(lambda g, key=key: (bw[key].copy_(g), None)[1])
        return out
    return wrapped


for name in TAPPED:
    if hasattr(ops, name):
        setattr(ops, name, tap(name, getattr(ops, name)))
_loss = tr.loss


def loss_reset(m, f):
    counter[0] = 0
    return _loss(m, f)


tr.loss = loss_reset

noise = []
if args.noise:
    ns = args.noise_shape or args.shape
    secs = 30 + args.replays * 0.004
    for _ in range(args.noise):
        noise.append(subprocess.Popen([sys.executable, os.path.abspath(__file__), "--as-noise", str(secs), "--shape", ns,
                                       "--batch", str(args.batch), "--streams", str(args.streams)] + (["--bf16"] if args.bf16 else [])))
    time.sleep(12)                                   # the noise processes import torch, capture and start replaying

if args.eager:
    tr._fwd_bwd(mov, fix)
    tr._fwd_bwd(mov, fix)
    step = lambda: tr._fwd_bwd(mov, fix)             # noqa: E731
else:
    tr.capture(mov, fix, verify=False)
    step = tr._graph.replay
torch.cuda.synchronize()
names = [n for n, _ in model.named_parameters()]
keys_f = list(order)
keys_b = [k for k in reversed(order) if k in bw]
taps = [("F " + k, fw[k]) for k in keys_f] + [("B " + k, bw[k]) for k in keys_b]
for n, (off, k) in zip(names, tr.fp.offsets):
    taps.append(("G " + n, tr.fp.grad[off:off + k]))
bufs = [t for _, t in taps]
step()
torch.cuda.synchronize()
refs = [t.clone() for t in bufs]
scale = torch.stack([r.abs().max() for r in refs]).clamp_min(1e-30)
print("%d taps (%d forward, %d backward, %d parameter gradients); %d noise processes" % (
    len(taps), len(keys_f), len(keys_b), len(names), len(noise)), flush=True)

rows = []
dumped = []
t0 = time.time()
for i in range(args.replays):
    step()
    poke()
    d = torch._foreach_sub(bufs, refs)
    rows.append(torch.stack(torch._foreach_norm(d, float("inf"))) / scale)
    if args.dump:
        e = rows[-1].cpu()
        hit = [j for j in range(len(taps)) if float(e[j]) > 1e-4]
        if any(not taps[j][0].startswith("G ") for j in hit) and len(dumped) < 4:
            dumped.append({"replay": i, "taps": {taps[j][0]: (refs[j].cpu(), bufs[j].cpu()) for j in hit}})
            if len(dumped) == 4:
                break
    if i % 200 == 199:
        torch.cuda.synchronize()
if args.dump:
    print("stopped after %d bad replays" % len(dumped))
    for dmp in dumped:
        name = [n for n in (t[0] for t in taps) if n in dmp["taps"] and not n.startswith("G ")][0]
        ref, badv = dmp["taps"][name]
        diff = (badv - ref).abs()
        m = float(ref.abs().max())
        idx = torch.nonzero(diff > 1e-3 * m)
        print("==== replay %d, first bad tap %s: %d of %d elements differ by > 1e-3 of max (max|ref| %.3e, max|diff| %.3e)" % (
            dmp["replay"], name, idx.shape[0], ref.numel(), m, float(diff.max())))
        if idx.numel():
            lo, hi = idx.min(0).values.tolist(), idx.max(0).values.tolist()
            print("     index box (b, z, y, x, c): lo %s hi %s" % (lo, hi))
            for d_ in range(1, idx.shape[1]):
                vals = sorted(set(idx[:, d_].tolist()))
                print("     dim %d values: %s" % (d_, vals[:40]))
            for r_ in idx[:12].tolist():
                t_ = tuple(r_)
                print("       %s  ref % .5e  got % .5e" % (t_, float(ref[t_]), float(badv[t_])))
            zeros = int((badv[diff > 1e-3 * m] == 0).sum())
            print("     of those, exactly zero in the bad run: %d; non-finite: %d" % (zeros, int((~torch.isfinite(badv)).sum())))
E = torch.stack(rows).cpu()                          # (replays, taps)
dt = time.time() - t0
alive = sum(p.poll() is None for p in noise)
print("%d replays in %.1f s (%d noise processes still running at the end)" % (args.replays, dt, alive), flush=True)
gsel = [j for j, (n, _) in enumerate(taps) if n.startswith("G ")]
gmax_scale = scale.cpu()[gsel].max()
gerr = (E[:, gsel] * scale.cpu()[gsel]).max(1).values / gmax_scale          # of the global max |g|, as the DP test measures
print("parameter gradients vs replay 0, of global max|g|: median %.2e  p99 %.2e  max %.2e" % (
    gerr.median(), gerr.kthvalue(max(1, int(0.99 * len(gerr)))).values, gerr.max()), flush=True)
floor = E.median(0).values
bad = torch.nonzero(gerr > args.thresh).flatten().tolist()
print("replays above %.0e: %d of %d %s" % (args.thresh, len(bad), args.replays, bad[:20]), flush=True)
from collections import Counter
first = Counter()
for i in bad:
    for j, (n, _) in enumerate(taps):
        if E[i, j] > max(20 * float(floor[j]), 1e-5):
            first[n] += 1
            break
print("first tap (execution order) that left its noise floor, over all bad replays:")
for n, c in first.most_common():
    print("   %3d x %s" % (c, n))
for i in bad[:3]:
    print("---- replay %d: parameter-gradient error %.3e of max|g|; taps that left their noise floor, in execution order:" % (i, gerr[i]))
    shown = 0
    for j, (n, _) in enumerate(taps):
        if E[i, j] > max(20 * float(floor[j]), 1e-5) and shown < 25:
            print("      %-64s err %.3e of own max (floor %.1e)" % (n, E[i, j], floor[j]))
            shown += 1
for p in noise:
    p.wait()
