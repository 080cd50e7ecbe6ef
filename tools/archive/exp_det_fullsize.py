"""deterministic mode at the benchmark shape: two eager steps bit-identical, and what the mode costs (the large feature warps take
the destination-tile kernels there: integer sums in LDS instead of 64-bit atomics on global memory)"""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from smilecode_amd import models, ops, synth  # noqa: E402
from smilecode_amd.engine import Trainer  # noqa: E402

shape = (160, 192, 160)
m = models.ModeT(shape, head_dim=6, num_heads=[8, 4, 2, 1, 1], scale=1).cuda()
models.load_numpy_weights(m, synth.make_weights(24))
mov, fix = (torch.from_numpy(a).cuda() for a in synth.make_pair(shape, 24))
res = {}
for det in (False, True):
    ops.set_deterministic(det)
    tr = Trainer(m)
    tr._fwd_bwd(mov, fix)
    g1 = tr.fp.grad.clone()
    tr._fwd_bwd(mov, fix)
    same = bool(torch.equal(tr.fp.grad, g1))
    tr.capture(mov, fix)
    for _ in range(3):
        tr._graph.replay()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(20):
        tr._graph.replay()
    torch.cuda.synchronize()
    ms = (time.perf_counter() - t0) / 20 * 1e3
    rep = bool(torch.equal(tr.fp.grad, g1))
    res[det] = g1
    print(f"deterministic={det}: two eager steps bit-identical {same}, graph replay == eager {rep}, fwd+bwd replay {ms:.3f} ms")
ops.set_deterministic(False)
print(f"max |grad diff| deterministic vs float atomics, of max |grad|: {float((res[True] - res[False]).abs().max() / res[False].abs().max()):.2e}")
