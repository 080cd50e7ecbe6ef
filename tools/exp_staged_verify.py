"""Single process: the staged backward (three autograd stages) eager and as three captured hipGraphs against the plain
eager step, PER PARAMETER TENSOR, under hostile allocator states:

  * garbage: a few GB of NaN / huge-value blocks are allocated, filled and freed right before every capture, so pool blocks
    handed to the capture hold hostile stale data;
  * poison: every torch.empty / empty_like on the GPU is filled with NaN -- in the eager staged step AND inside the capture
    (the fill kernels become graph nodes, so every replay re-poisons its buffers): a read of memory nobody wrote is NaN.

    python tools/exp_staged_verify.py [shape] [--rounds N] [--bf16]
"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from smilecode_amd import models, synth                      # noqa: E402
from smilecode_amd.engine import Trainer                      # noqa: E402

shape = (32, 48, 32)
rounds, bf16, batch = 4, False, 1
argv = sys.argv[1:]
while argv:
    a = argv.pop(0)
    if a == "--rounds":
        rounds = int(argv.pop(0))
    elif a == "--bf16":
        bf16 = True
    elif a == "--batch":
        batch = int(argv.pop(0))
    else:
        shape = tuple(int(s) for s in a.split(","))
dev = torch.device("cuda")
_empty, _empty_like = torch.empty, torch.empty_like


def p_empty(*a, **k):
    t = _empty(*a, **k)
    if t.is_cuda and t.is_floating_point():
        t.fill_(float("nan"))
    return t


def p_empty_like(*a, **k):
    t = _empty_like(*a, **k)
    if t.is_cuda and t.is_floating_point():
        t.fill_(float("nan"))
    return t


class poisoned:
    def __enter__(self):
        torch.empty, torch.empty_like = p_empty, p_empty_like

    def __exit__(self, *exc):
        torch.empty, torch.empty_like = _empty, _empty_like


def make(overlap):
    kw = dict(act_dtype=torch.bfloat16) if bf16 else {}
    model = models.ModeT(shape, head_dim=6, num_heads=[8, 4, 2, 1, 1], scale=1, **kw).to(dev)
    models.load_numpy_weights(model, synth.make_weights(24))
    return Trainer(model, overlap_allreduce=overlap)


mov, fix = (torch.from_numpy(a).to(dev) for a in synth.make_pair(shape, 24, batch))
plain = make(False)
plain._fwd_bwd(mov, fix)
plain._fwd_bwd(mov, fix)
torch.cuda.synchronize()
ref = plain.fp.grad.clone()
plain._fwd_bwd(mov, fix)
torch.cuda.synchronize()
names = [n for n, _ in plain.model.named_parameters()]
gmax = float(ref.abs().max())
print("plain eager step run to run: %.3e of max|g|" % (float((plain.fp.grad - ref).abs().max()) / gmax), flush=True)


def report(tag, tr):
    torch.cuda.synchronize()
    g = tr.fp.grad
    rows = []
    for n, (off, k) in zip(names, tr.fp.offsets):
        a, b = ref[off:off + k], g[off:off + k]
        if not bool(torch.isfinite(b).all()):
            rows.append((float("inf"), float("inf"), n))
            continue
        d = float((a - b).abs().max())
        rows.append((d / gmax, d / max(float(a.abs().max()), 1e-30), n))
    rows.sort(reverse=True)
    bad = [r for r in rows if not r[0] <= 5e-6]
    print("%-46s worst %.3e of max|g| (%s)%s" % (tag, rows[0][0], rows[0][2], "" if not bad else "   <-- %d tensors > 5e-6" % len(bad)),
          flush=True)
    for a, b, n in bad[:10]:
        print("        %-36s %.3e of global max, %.3e of own max" % (n, a, b), flush=True)
    return rows[0][0]


def garbage(gb, val):
    blocks = [_empty(int(s), dtype=torch.float32, device=dev).fill_(val) for s in (gb * 2 ** 28 * f for f in (0.5, 0.25, 0.125, 0.125))]
    small = [_empty(int(n), dtype=torch.float32, device=dev).fill_(val) for n in (1 << 12, 1 << 14, 1 << 16, 1 << 18, 1 << 20) for _ in range(24)]
    torch.cuda.synchronize()
    del blocks, small


worst = 0.0
for r in range(rounds):
    tr = make(True)
    tr._fwd_bwd_staged(mov, fix)
    tr._fwd_bwd_staged(mov, fix)
    worst = max(worst, report("round %d eager staged" % r, tr))
    with poisoned():
        tr._fwd_bwd_staged(mov, fix)
    worst = max(worst, report("round %d eager staged, poisoned empty()" % r, tr))
    garbage(2, float("nan") if r % 2 == 0 else 3e30)
    if r % 2 == 1:
        torch.cuda.empty_cache()
    tr.capture(mov, fix, verify=False)
    for rep in range(3):
        tr.fp.grad.fill_(float("nan"))
        for gr in tr._stage_graphs:
            gr.replay()
        worst = max(worst, report("round %d captured staged replay %d" % (r, rep), tr))
    # parameters move, replays must follow (the packed weights are part of the step)
    tr.release_graph()
    del tr
    tr = make(True)
    garbage(1, float("nan"))
    with poisoned():
        tr.capture(mov, fix, verify=False)
    for rep in range(2):
        tr.fp.grad.fill_(float("nan"))
        for gr in tr._stage_graphs:
            gr.replay()
        worst = max(worst, report("round %d captured under poison, replay %d" % (r, rep), tr))
    tr.release_graph()
    del tr
print("worst over everything: %.3e" % worst)
