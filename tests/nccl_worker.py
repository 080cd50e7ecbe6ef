"""Worker for tests/test_gpu_e2e.py::test_staged_step_through_the_nccl_backend: ONE rank on the `nccl` backend (= RCCL), the
overlapped step with its three bucket all-reduces actually issued (``always_reduce``): RCCL's collectives are STREAM-ORDERED
work on RCCL's own stream (gloo stages through pinned host memory on copy streams), so this is the ordering the 8-GPU run has
-- event after stage k -> all-reduce of bucket k beside stage k + 1 -> join before Adam -- on the hardware we do have."""
import os
import sys

import numpy as np
import torch
import torch.distributed as dist

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from smilecode_amd import models, synth                       # noqa: E402
from smilecode_amd.engine import Trainer                       # noqa: E402

out_dir, shape, graph, steps = sys.argv[1], tuple(int(s) for s in sys.argv[2].split(",")), sys.argv[3] == "1", int(sys.argv[4])
os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
torch.cuda.set_device(0)
dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda", 0))
model = models.ModeT(shape, head_dim=6, num_heads=[8, 4, 2, 1, 1], scale=1).cuda()
models.load_numpy_weights(model, synth.make_weights(24))
tr = Trainer(model, overlap_allreduce=True)
tr.buckets.always_reduce = True
mov, fix = (torch.from_numpy(a).cuda() for a in synth.make_pair(shape, 24, 1))
if graph:
    tr.capture(mov, fix)
    assert tr._stage_graphs is not None
losses, grads = [], []
for _ in range(steps):
    out = tr.train_step(mov, fix, epoch=0)
    grads.append(tr.fp.grad.clone())
    losses.append(torch.stack([o.reshape(()) for o in out]))
torch.cuda.synchronize()
np.savez(os.path.join(out_dir, "nccl.npz"), flat=tr.fp.flat.cpu().numpy(), grads=torch.stack(grads).cpu().numpy(),
         losses=torch.stack(losses).cpu().numpy(), n_collectives=np.int64(tr.buckets.n_launched))
dist.destroy_process_group()
