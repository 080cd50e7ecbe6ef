import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from smilecode_amd import models, ops, synth
from smilecode_amd.engine import Trainer
shape = tuple(int(s) for s in (sys.argv[1] if len(sys.argv) > 1 else "160,192,160").split(","))
dev = torch.device("cuda")
model = models.ModeT(shape, head_dim=6, num_heads=[8, 4, 2, 1, 1], scale=1).to(dev)
models.load_numpy_weights(model, synth.make_weights(24))
mov, fix = (torch.from_numpy(a).to(dev) for a in synth.make_pair(shape, 24, 1))
tr = Trainer(model, lr=1e-4, max_epoch=30, weights=[1, 1])
names = [n for n, _ in model.named_parameters()]
ref = None
for rep in range(4):
    tr.fp.grad.fill_(float("nan"))
    tr._fwd_bwd(mov, fix)
    torch.cuda.synchronize()
    g = tr.fp.grad.clone()
    nn = [n for n, (off, k) in zip(names, tr.fp.offsets) if not bool(torch.isfinite(g[off:off + k]).all())]
    big = float(g[torch.isfinite(g)].abs().max())
    d = None if ref is None else float((g - ref).abs().max())
    print("eager pass %d: non-finite tensors %d, |grad| max %.3e, diff to pass 0 %s" % (rep, len(nn), big, d), nn[:4])
    if ref is None: ref = g
