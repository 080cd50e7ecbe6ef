import os, sys, torch, traceback
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from smilecode_amd import models, synth, ops
from smilecode_amd.engine import Trainer
shape = (32, 48, 32)
model = models.ModeT(shape, head_dim=6, num_heads=[8, 4, 2, 1, 1], scale=1).cuda()
models.load_numpy_weights(model, synth.make_weights(24))
mov, fix = (torch.from_numpy(a).cuda() for a in synth.make_pair(shape, 24))
tr = Trainer(model, overlap_allreduce=True)
loss, sim, reg = tr.loss(mov, fix)
M, Fx = model.last_features
pooled = model.encoder.last_pooled
def walk(fn, depth=0, seen=None, maxd=6):
    seen = seen if seen is not None else set()
    if fn is None or id(fn) in seen or depth > maxd: return
    seen.add(id(fn))
    print("  " * depth + type(fn).__name__)
    for nf, _ in fn.next_functions:
        walk(nf, depth + 1, seen, maxd)
print("grad_fn of M[4]:"); walk(M[4].grad_fn)
# stage 0 on the level-5 cuts only
g = torch.autograd.grad(loss, [M[4], Fx[4]], retain_graph=True)
print("stage 0 (level-5 cuts only, retain) ok")
try:
    g2 = torch.autograd.grad([M[4], Fx[4]], [pooled[3]], list(g), retain_graph=True)
    print("stage 1 from level-5 cuts to its pooled input ok", tuple(g2[0].shape))
except Exception as e:
    traceback.print_exc()
