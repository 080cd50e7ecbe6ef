"""Which backward intermediate goes wrong from the second replay of the captured train step?  retain_grad() on module outputs
during capture; after each replay the retained gradients are compared with those of replay 0."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from smilecode_amd import models, ops, synth
from smilecode_amd.engine import Trainer
shape = (64, 64, 64)
dev = torch.device("cuda")
model = models.ModeT(shape, head_dim=6, num_heads=[8, 4, 2, 1, 1], scale=1).to(dev)
models.load_numpy_weights(model, synth.make_weights(24))
mov, fix = (torch.from_numpy(a).to(dev) for a in synth.make_pair(shape, 24, 1))
kept = {}
order = []
def hook(name):
    def f(mod, inp, out):
        outs = out if isinstance(out, (tuple, list)) else [out]
        flat = []
        for o in outs:
            flat += list(o) if isinstance(o, (tuple, list)) else [o]
        for i, o in enumerate(flat):
            if torch.is_tensor(o) and o.requires_grad and o.grad_fn is not None:
                o.retain_grad()
                key = f"{name}[{i}]"
                if key not in kept: order.append(key)
                kept[key] = o
    return f
for name, mod in model.named_modules():
    if name and name.count(".") <= 1:
        mod.register_forward_hook(hook(name))
tr = Trainer(model, lr=1e-4, max_epoch=30, weights=[1, 1])
tr.capture(mov, fix)
snaps = []
for rep in range(3):
    tr._graph.replay()
    torch.cuda.synchronize()
    snaps.append({k: (kept[k].grad.clone() if kept[k].grad is not None else None) for k in order})
for k in order:
    a, b = snaps[0][k], snaps[1][k]
    if a is None: continue
    d = float((a - b).abs().max()); m = float(a.abs().max())
    flag = "  <-- differs" if not (d <= 1e-4 * m + 1e-12) else ""
    print("%-34s shape %-26s |grad| max %.3e   replay1 - replay0 %.3e%s" % (k, tuple(a.shape), m, d, flag))
