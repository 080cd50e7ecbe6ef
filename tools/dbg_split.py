import os, sys, torch
sys.path.insert(0, "/root/repo")
from smilecode_amd import ops, models, synth, _lib
shape = (32, 48, 32)
m = models.ModeT(shape, head_dim=6, num_heads=[8, 4, 2, 1, 1], scale=1).cuda()
models.load_numpy_weights(m, synth.make_weights(24))
from smilecode_amd.engine import Trainer
tr = Trainer(m)
L = _lib.load()
for name in ("modet_conv3d_fwd_stats", "modet_conv3d_fwd", "modet_conv3d_bwd_data", "modet_instnorm_lrelu_fwd_stats", "modet_conv3d_bwd_weight"):
    orig = getattr(L, name)
    def wrap(*a, _o=orig, _n=name):
        print("call", _n, [x for x in a if isinstance(x, int) and x < 100000][-8:], "ptr-align", [hex(x & 15) for x in a[:4] if isinstance(x, int) and x > 1 << 20], flush=True)
        r = _o(*a)
        torch.cuda.synchronize()
        return r
    setattr(L, name, wrap)
mov, fix = (torch.from_numpy(a).cuda() for a in synth.make_pair(shape, 24))
loss, sim, reg = tr.loss(mov, fix)
torch.cuda.synchronize()
print("forward ok", float(loss))
loss.backward()
torch.cuda.synchronize()
print("backward ok")
