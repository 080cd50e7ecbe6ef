import sys, time, torch
sys.path.insert(0, "/root/repo")
from smilecode_amd import models, synth
from smilecode_amd.engine import Trainer
shape = tuple(int(s) for s in (sys.argv[1] if len(sys.argv) > 1 else "32,48,32").split(","))
def mk():
    m = models.ModeT(shape, head_dim=6, num_heads=[8,4,2,1,1], scale=1).cuda()
    models.load_numpy_weights(m, synth.make_weights(24))
    return Trainer(m)
mov, fix = synth.make_pair(shape, 24)
mov, fix = torch.from_numpy(mov).cuda(), torch.from_numpy(fix).cuda()
a, b = mk(), mk()
b.capture(mov, fix)
for i in range(3):
    la = a.train_step(mov, fix)
    lb = b.train_step(mov, fix)
    print(i, float(la[0]), float(lb[0]), float((a.fp.flat - b.fp.flat).abs().max()), float((a.fp.grad - b.fp.grad).abs().max() / a.fp.grad.abs().max()))
for t, name in ((a, "eager"), (b, "graph")):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(20): t.train_step(mov, fix)
    host = time.perf_counter() - t0
    torch.cuda.synchronize(); dt = time.perf_counter() - t0
    print(name, "ms/step", dt / 20 * 1e3, "host enqueue ms/step", host / 20 * 1e3)
