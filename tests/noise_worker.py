"""Worker for tests/test_gpu_e2e.py::test_attention_backward_is_bit_stable_beside_another_process: keeps the GPU busy the way a
second rank on the same device does -- the captured train step replayed in a loop, plus tiny kernels on a few high- and
normal-priority streams (torch.distributed's gloo CUDA path opens such a pool) -- for argv[1] seconds."""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from smilecode_amd import models, synth                       # noqa: E402
from smilecode_amd.engine import Trainer                       # noqa: E402

seconds, shape = float(sys.argv[1]), (32, 48, 32)
model = models.ModeT(shape, head_dim=6, num_heads=[8, 4, 2, 1, 1], scale=1).cuda()
models.load_numpy_weights(model, synth.make_weights(24))
mov, fix = (torch.from_numpy(a).cuda() for a in synth.make_pair(shape, 24, 1))
tr = Trainer(model).capture(mov, fix, verify=False)
streams = [torch.cuda.Stream(priority=-1) for _ in range(4)] + [torch.cuda.Stream() for _ in range(4)]
ticks = [torch.zeros(256, device="cuda") for _ in streams]
open(sys.argv[2], "w").write("ready")
t0 = time.time()
while time.time() - t0 < seconds:
    for _ in range(20):
        tr._graph.replay()
        for st, t in zip(streams, ticks):
            with torch.cuda.stream(st):
                t.add_(1.0)
    torch.cuda.synchronize()
