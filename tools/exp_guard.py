"""Out-of-bounds hunt without a GPU sanitizer: every torch.empty / empty_like / zeros(_like) the package issues on the GPU
gets a NaN-filled GUARD BAND on both sides (and a NaN interior for empty), all buffers are kept alive for the step, then

  * an out-of-bounds (or uninitialised) READ whose value is used shows up as NaN in the losses / gradients;
  * an out-of-bounds WRITE shows up as a guard band that is no longer all-NaN (the buffer is named by its allocation site).

    python tools/exp_guard.py [shape] [--batch B] [--staged] [--bf16] [--guard ELEMS]
"""
import os
import sys
import traceback

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from smilecode_amd import models, synth                      # noqa: E402
from smilecode_amd.engine import Trainer                      # noqa: E402

shape, batch, staged, bf16, G, from_start = (32, 48, 32), 1, False, False, 4096, False
argv = sys.argv[1:]
while argv:
    a = argv.pop(0)
    if a == "--batch":
        batch = int(argv.pop(0))
    elif a == "--staged":
        staged = True
    elif a == "--bf16":
        bf16 = True
    elif a == "--from-start":      # guard + poison from the FIRST pass on: the recording pass, the packed-weights arena
        from_start = True
    elif a == "--guard":
        G = int(argv.pop(0))
    else:
        shape = tuple(int(s) for s in a.split(","))
dev = torch.device("cuda")
_empty, _empty_like, _zeros, _zeros_like = torch.empty, torch.empty_like, torch.zeros, torch.zeros_like
live = []          # (site, whole buffer, n)


def _site():
    for fr in traceback.extract_stack()[:-3][::-1]:
        if "smilecode_amd" in fr.filename:
            return "%s:%d %s" % (os.path.basename(fr.filename), fr.lineno, fr.name)
    return "?"


def _guarded(shape_, dtype, device, fill):
    if isinstance(shape_, int):
        shape_ = (shape_,)
    n = 1
    for s in shape_:
        n *= int(s)
    whole = _empty(n + 2 * G, dtype=dtype, device=device)
    whole.fill_(float("nan"))
    inner = whole[G:G + n]
    if fill is not None:
        inner.fill_(fill)
    live.append((_site(), whole, n))
    return inner.view(tuple(shape_))


def _norm_shape(a):
    if len(a) == 1 and isinstance(a[0], (tuple, list, torch.Size)):
        return tuple(a[0])
    return tuple(a)


def g_empty(*a, **k):
    dt, dv = k.get("dtype", torch.float32), k.get("device", None)
    if dv is not None and torch.device(dv).type == "cuda" and dt in (torch.float32, torch.bfloat16) and len(k.keys() - {"dtype", "device"}) == 0:
        return _guarded(_norm_shape(a), dt, dv, None)
    return _empty(*a, **k)


def g_zeros(*a, **k):
    dt, dv = k.get("dtype", torch.float32), k.get("device", None)
    if dv is not None and torch.device(dv).type == "cuda" and dt in (torch.float32, torch.bfloat16) and len(k.keys() - {"dtype", "device"}) == 0:
        return _guarded(_norm_shape(a), dt, dv, 0.0)
    return _zeros(*a, **k)


def g_empty_like(t, **k):
    if t.is_cuda and t.dtype in (torch.float32, torch.bfloat16) and not k and t.is_contiguous():
        return _guarded(tuple(t.shape), t.dtype, t.device, None)
    return _empty_like(t, **k)


def g_zeros_like(t, **k):
    if t.is_cuda and t.dtype in (torch.float32, torch.bfloat16) and not k and t.is_contiguous():
        return _guarded(tuple(t.shape), t.dtype, t.device, 0.0)
    return _zeros_like(t, **k)


kw = dict(act_dtype=torch.bfloat16) if bf16 else {}
model = models.ModeT(shape, head_dim=6, num_heads=[8, 4, 2, 1, 1], scale=1, **kw).to(dev)
models.load_numpy_weights(model, synth.make_weights(24))
mov, fix = (torch.from_numpy(a).to(dev) for a in synth.make_pair(shape, 24, batch))
tr = Trainer(model, overlap_allreduce=staged)
run = tr._fwd_bwd_staged if staged else tr._fwd_bwd
run(mov, fix)
run(mov, fix)
torch.cuda.synchronize()
ref = tr.fp.grad.clone()
names = [n for n, _ in model.named_parameters()]
if from_start:
    model2 = models.ModeT(shape, head_dim=6, num_heads=[8, 4, 2, 1, 1], scale=1, **kw).to(dev)
    models.load_numpy_weights(model2, synth.make_weights(24))
    tr = Trainer(model2, overlap_allreduce=staged)            # fresh step context: nothing recorded, no arena yet
    run = tr._fwd_bwd_staged if staged else tr._fwd_bwd

torch.empty, torch.empty_like, torch.zeros, torch.zeros_like = g_empty, g_empty_like, g_zeros, g_zeros_like
try:
    npass = 3 if from_start else 1
    for i in range(npass):
        out = run(mov, fix)
        torch.cuda.synchronize()
        if i + 1 < npass:
            g = tr.fp.grad
            fin = bool(torch.isfinite(g).all())
            print("pass %d: finite %s, max diff vs reference %.3e of max|g|" % (
                i, fin, float((g - ref).abs().max() / ref.abs().max()) if fin else float("nan")))
finally:
    torch.empty, torch.empty_like, torch.zeros, torch.zeros_like = _empty, _empty_like, _zeros, _zeros_like
print("shape %s batch %d staged %s bf16 %s: %d guarded buffers, guard %d elements" % (shape, batch, staged, bf16, len(live), G))
print("losses:", [float(v) for v in out])
g = tr.fp.grad
gmax = float(ref.abs().max())
bad = 0
for n, (off, k) in zip(names, tr.fp.offsets):
    a, b = ref[off:off + k], g[off:off + k]
    fin = bool(torch.isfinite(b).all())
    d = float((a - b).abs().max()) if fin else float("inf")
    if not d <= 5e-6 * gmax:
        bad += 1
        print("  READ?  %-36s diff %.3e of max|g|%s" % (n, d / gmax, "" if fin else " (non-finite)"))
print("parameter tensors that differ from the unguarded step:", bad)
nw = 0
for site, whole, n in live:
    lo, hi = whole[:G], whole[G + n:]
    if not (bool(torch.isnan(lo.float()).all()) and bool(torch.isnan(hi.float()).all())):
        nw += 1
        blo = torch.nonzero(~torch.isnan(lo.float())).flatten()
        bhi = torch.nonzero(~torch.isnan(hi.float())).flatten()
        print("  WRITE  %s (%d elements, %s): %d guard elements below (nearest %s), %d above (first %s)" % (
            site, n, str(whole.dtype), blo.numel(), (G - int(blo.max())) if blo.numel() else "-", bhi.numel(),
            int(bhi.min()) if bhi.numel() else "-"))
print("buffers with a damaged guard band:", nw)
