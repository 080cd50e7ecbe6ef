"""One rank of the overlapped-step stress (launched by tools/dp_stress.py through torch.distributed.run, N ranks on ONE GPU over
gloo, or one rank over nccl): MANY train steps per process with lr = 0 -- the parameters never move, so every step must return
the same gradients up to the float-atomic noise -- with the gradient checked ON DEVICE after every step, before and after the
all-reduce.  Names the step, the bucket (= stage) and the side (local replay / reduced) of anything that leaves the noise floor.

Variations through the environment (one hypothesis each):
  STRESS_NO_INPUT_COPY=1   skip the refresh of the static input buffers in front of graph 0
  STRESS_NO_ALLREDUCE=1    no collective between the replays
  STRESS_SYNC_STAGES=1     torch.cuda.synchronize() after every stage graph (no overlap of anything)
  STRESS_NO_ADAM=1         no optimizer kernel after the step
  STRESS_LOCAL_SNAP=0      do not snapshot the local gradients between replay and all-reduce (no extra kernels in the step)
  STRESS_EAGER=1           eager stages instead of graphs
"""
import os
import sys

import torch
import torch.distributed as dist

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from smilecode_amd import models, ops, synth                  # noqa: E402
from smilecode_amd.engine import Trainer                       # noqa: E402
from smilecode_amd.parallel import init_from_env               # noqa: E402

shape = tuple(int(s) for s in sys.argv[1].split(","))
steps = int(sys.argv[2])
E = os.environ.get
backend = E("STRESS_BACKEND", "gloo")
rank, local, world = init_from_env(backend)
if world == 1 and backend == "nccl":
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    os.environ.setdefault("MASTER_PORT", "29555")
    dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda", 0))
torch.cuda.set_device(0)
model = models.ModeT(shape, head_dim=6, num_heads=[8, 4, 2, 1, 1], scale=1).cuda()
models.load_numpy_weights(model, synth.make_weights(24))
tr = Trainer(model, lr=0.0, overlap_allreduce=True)
if world == 1:
    tr.buckets.always_reduce = True
mov, fix = synth.make_pair(shape, 24, max(world, 2))
mov, fix = torch.from_numpy(mov[rank:rank + 1]).cuda(), torch.from_numpy(fix[rank:rank + 1]).cuda()
eager = E("STRESS_EAGER") == "1"
if not eager:
    tr.capture(mov, fix)
flat0 = tr.fp.flat.clone()
names = [n for n, _ in model.named_parameters()]
B = tr.buckets
snap = E("STRESS_LOCAL_SNAP", "1") == "1"
loc = [torch.zeros(b - a, device="cuda") for a, b in B.ranges]


def step():
    if E("STRESS_NO_INPUT_COPY") != "1" and not eager:
        tr._static_in[0].copy_(mov, non_blocking=True)
        tr._static_in[1].copy_(fix, non_blocking=True)
    B.begin_staged()

    def after(k):
        if snap:
            a, b = B.ranges[k]
            loc[k].copy_(tr.fp.grad[a:b])
        if E("STRESS_SYNC_STAGES") == "1":
            torch.cuda.synchronize()
        if E("STRESS_NO_ALLREDUCE") != "1":
            B.launch(k)
    if eager:
        tr._fwd_bwd_staged(mov, fix, after)
    else:
        for k, gr in enumerate(tr._stage_graphs):
            gr.replay()
            after(k)
    scale = B.finish_staged()
    if E("STRESS_NO_ADAM") != "1":
        tr.step += 1
        ops.adam_amsgrad_step_(tr.fp.flat, tr.fp.grad, tr.m, tr.v, tr.vmax, 0.0, tr.step, 0.9, 0.999, 1e-8, scale)


step()
torch.cuda.synchronize()
ref_red = tr.fp.grad.clone()
ref_loc = [t.clone() for t in loc]
gmax = ref_red.abs().max()
lmax = torch.stack([t.abs().max() for t in ref_loc]).max()
errs = []
import time
t0 = time.time()
for i in range(steps):
    step()
    e_red = torch.stack([(tr.fp.grad[a:b] - ref_red[a:b]).abs().max() for a, b in B.ranges]) / gmax
    e_loc = torch.stack([(loc[k] - ref_loc[k]).abs().max() for k in range(3)]) / lmax
    errs.append(torch.cat([e_red, e_loc]))
    if i % 50 == 49:
        torch.cuda.synchronize()
torch.cuda.synchronize()
Eall = torch.stack(errs).cpu()
moved = float((tr.fp.flat - flat0).abs().max())
bad = torch.nonzero(Eall.max(1).values > 2e-5).flatten().tolist()
print("rank %d: %.1f ms per step" % (rank, (time.time() - t0) / steps * 1e3))
print("rank %d: %d steps, parameters moved by %.1e; reduced-gradient error vs step 0: median %.2e max %.2e; local: median %.2e max %.2e; steps above 2e-5: %d %s" % (
    rank, steps, moved, Eall[:, :3].max(1).values.median(), Eall[:, :3].max(), Eall[:, 3:].max(1).values.median(), Eall[:, 3:].max(), len(bad), bad[:12]),
    flush=True)
for i in bad[:8]:
    print("rank %d   step %d: reduced per bucket %s | local per bucket %s" % (
        rank, i, ["%.2e" % v for v in Eall[i, :3]], ["%.2e" % v for v in Eall[i, 3:]]), flush=True)
if dist.is_initialized():
    dist.barrier()
    dist.destroy_process_group()
