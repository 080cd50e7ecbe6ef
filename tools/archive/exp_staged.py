"""staged backward (engine.Trainer, overlapped all-reduce) == plain backward, eager and as three hipGraphs"""
import os, sys, torch, traceback
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from smilecode_amd import models, synth, ops
from smilecode_amd.engine import Trainer
shape = (32, 48, 32)
model = models.ModeT(shape, head_dim=6, num_heads=[8, 4, 2, 1, 1], scale=1).cuda()
models.load_numpy_weights(model, synth.make_weights(24))
mov, fix = (torch.from_numpy(a).cuda() for a in synth.make_pair(shape, 24))
ref = Trainer(model)
ref._fwd_bwd(mov, fix)
gref = ref.fp.grad.clone()
tr = Trainer(model, overlap_allreduce=True)
tr._fwd_bwd_staged(mov, fix)
torch.cuda.synchronize()
print("eager staged: max |staged - plain| %.3e at scale %.3e" % (float((tr.fp.grad - gref).abs().max()), float(gref.abs().max())))
tr.capture(mov, fix)
tr.fp.grad.zero_()
for g in tr._stage_graphs:
    g.replay()
torch.cuda.synchronize()
print("three graphs: max |staged - plain| %.3e" % float((tr.fp.grad - gref).abs().max()))
