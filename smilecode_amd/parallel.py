"""Data-parallel plumbing for the ModeT train step: one process per GPU, torch.distributed
(backend "nccl" = RCCL over xGMI on ROCm; "gloo" in the CPU tests).

The reference is single-GPU (train.py:183-189).  Volume pairs are independent units
(InstanceNorm is per sample), so rank r takes pairs r, r+world, ... and the only
exchange is ONE all-reduce of the flat 4.12 MB gradient buffer per step; its 1/world scale is
folded into the fused Adam kernel.  Host-side only: no kernels here."""
from __future__ import annotations

import os

import torch
import torch.distributed as dist


def init_from_env(backend=None):
    """torchrun-style rendezvous (RANK / LOCAL_RANK / WORLD_SIZE / MASTER_ADDR / MASTER_PORT).
    Returns (rank, local_rank, world).  A single process needs no process group."""
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if world > 1 and not dist.is_initialized():
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29500")
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        if backend is None:
            backend = os.environ.get("MODET_DIST_BACKEND") or ("nccl" if torch.cuda.is_available() else "gloo")
        if backend == "nccl":
            torch.cuda.set_device(local)
            dist.init_process_group(backend, rank=rank, world_size=world, device_id=torch.device("cuda", local))
        else:
            dist.init_process_group(backend, rank=rank, world_size=world)
    if torch.cuda.is_available() and local >= torch.cuda.device_count():
        local = local % torch.cuda.device_count()       # several ranks on one GPU (gloo smoke tests only)
    return rank, local, world


def pairs_for_rank(n_pairs: int, rank: int, world: int):
    """indices of the volume pairs rank `rank` processes (round robin, no data-path collective)"""
    return list(range(rank, n_pairs, world))


def lockstep_pairs_for_rank(n_pairs: int, rank: int, world: int):
    """round-robin shard with the SAME number of entries on every rank: when ``n_pairs % world != 0`` the list wraps
    around (as torch's DistributedSampler pads), so every rank runs ``ceil(n_pairs / world)`` steps.  Each step holds a
    blocking gradient all-reduce; ranks with different step counts would pair all-reduces of different steps (and of
    different epochs' learning rates) and hang at the end of training."""
    per_rank = (n_pairs + world - 1) // world
    return [(rank + i * world) % n_pairs for i in range(per_rank)]


class FlatParams:
    """All parameters (and their gradients) of a module as views into two flat fp32 buffers, so the
    gradient all-reduce is one collective and the optimizer is one kernel."""

    def __init__(self, module: torch.nn.Module):
        params = [p for p in module.parameters() if p.requires_grad]
        self.params = params
        n = sum(p.numel() for p in params)
        dev = params[0].device
        self.flat = torch.empty(n, dtype=torch.float32, device=dev)
        self.grad = torch.zeros(n, dtype=torch.float32, device=dev)
        off = 0
        self.offsets = []
        with torch.no_grad():
            for p in params:
                k = p.numel()
                self.flat[off:off + k].copy_(p.detach().reshape(-1))
                p.data = self.flat[off:off + k].view(p.shape)
                p.grad = self.grad[off:off + k].view(p.shape)
                self.offsets.append((off, k))
                off += k
        self.numel = n

    def zero_grad(self):
        """Detach the per-parameter .grad tensors: autograd then *assigns* the first incoming gradient instead of
        launching one tiny add kernel per parameter into a pre-zeroed buffer; `gather_grads` packs them afterwards."""
        for p in self.params:
            p.grad = None

    def segment_index(self):
        """int64 tensor (numel): element of the flat buffers -> index of its parameter in ``params``"""
        if getattr(self, "_seg", None) is None:
            sizes = torch.tensor([k for _, k in self.offsets], device=self.flat.device)
            self._seg = torch.repeat_interleave(torch.arange(len(self.offsets), device=self.flat.device), sizes)
        return self._seg

    def grad_destinations(self):
        """parameter.data_ptr() -> that parameter's view of the flat gradient buffer (ops.deferred_wgrad_reductions)"""
        if getattr(self, "_dst", None) is None:
            self._dst = {p.data_ptr(): self.grad[off:off + k].view(p.shape) for p, (off, k) in zip(self.params, self.offsets)}
        return self._dst

    def gather_grads(self, written=()):
        """pack every parameter gradient into the flat buffer (one fused multi-tensor copy); parameters whose pointer is
        in ``written`` already have theirs there (written directly by the deferred weight-gradient reductions)"""
        views, srcs, missing = [], [], []
        for p, (off, k) in zip(self.params, self.offsets):
            if p.data_ptr() in written:
                if p.grad is not None:            # a further use of the parameter went through autograd: add it on top
                    self.grad[off:off + k].view(p.shape).add_(p.grad)
                continue
            v = self.grad[off:off + k].view(p.shape)
            if p.grad is None:
                missing.append(v)
            else:
                views.append(v)
                srcs.append(p.grad)
        if views:
            torch._foreach_copy_(views, srcs)
        if missing:                           # parameters no loss term reaches: one multi-tensor launch, not one fill each
            torch._foreach_zero_(missing)

    def allreduce_grads(self, group=None):
        """sum over ranks (in place); returns the factor the optimizer must apply (1/world)"""
        if dist.is_available() and dist.is_initialized() and dist.get_world_size(group) > 1:
            dist.all_reduce(self.grad, op=dist.ReduceOp.SUM, group=group)
            return 1.0 / dist.get_world_size(group)
        return 1.0


def broadcast_parameters(flat: FlatParams, src=0, group=None):
    if dist.is_available() and dist.is_initialized() and dist.get_world_size(group) > 1:
        dist.broadcast(flat.flat, src=src, group=group)


class BucketedAllReduce:
    """Gradient all-reduce in a few buckets launched from backward hooks, so the collective overlaps the rest of the
    backward pass (BASELINE.json configs[4] "overlapped all-reduce").

    Buckets are contiguous ranges of the flat gradient buffer, given as lists of parameter-name prefixes in the order
    they become READY during backward (ModeT: the per-level projection / attention / CWM parameters first, then the coarse
    encoder levels -- 86 % of all parameters -- and the full-resolution encoder blocks last).  A bucket fires when every
    one of its parameters has received its gradient: one fused copy into the flat range, then
    ``all_reduce(range, async_op=True)``: torch.distributed's RCCL stream waits for the copy and runs beside the
    remaining backward kernels; ``finish()`` joins the streams before the optimizer kernel.  Parameters that never receive
    a gradient are zero-filled and their bucket is reduced in ``finish()``."""

    def __init__(self, flat: FlatParams, named_parameters, bucket_prefixes, group=None):
        self.flat, self.group = flat, group
        names = [n for n, p in named_parameters if p.requires_grad]
        if len(names) != len(flat.params):
            raise RuntimeError("BucketedAllReduce: named_parameters does not match the flat buffer")
        self.bucket_of = []
        for n in names:
            hit = [i for i, pre in enumerate(bucket_prefixes) if any(n.startswith(q) for q in pre)]
            if len(hit) != 1:
                raise RuntimeError(f"BucketedAllReduce: parameter {n} matches {len(hit)} buckets")
            self.bucket_of.append(hit[0])
        self.nb = len(bucket_prefixes)
        self.members = [[i for i, b in enumerate(self.bucket_of) if b == k] for k in range(self.nb)]
        for k, mem in enumerate(self.members):
            if not mem or mem != list(range(mem[0], mem[-1] + 1)):
                raise RuntimeError(f"BucketedAllReduce: bucket {k} is empty or not contiguous in the flat buffer")
        self.ranges = [(flat.offsets[m[0]][0], flat.offsets[m[-1]][0] + flat.offsets[m[-1]][1]) for m in self.members]
        self.pending, self.works, self.fired = [0] * self.nb, [], [False] * self.nb
        self.enabled = False
        self.always_reduce = False      # issue the collectives in a one-rank group too (tests: RCCL's stream ordering at world size 1)
        self.n_launched = 0
        for i, prm in enumerate(flat.params):
            prm.register_post_accumulate_grad_hook(self._make_hook(i))

    def _make_hook(self, i):
        def hook(param):
            if not self.enabled:
                return
            k = self.bucket_of[i]
            self.pending[k] -= 1
            if self.pending[k] == 0:
                self._fire(k)
        return hook

    def _fire(self, k):
        views, srcs = [], []
        for i in self.members[k]:
            off, n = self.flat.offsets[i]
            v = self.flat.grad[off:off + n].view(self.flat.params[i].shape)
            g = self.flat.params[i].grad
            if g is None:
                v.zero_()
            else:
                views.append(v)
                srcs.append(g)
        if views:
            torch._foreach_copy_(views, srcs)
        a, b = self.ranges[k]
        self.fired[k] = True
        if dist.is_available() and dist.is_initialized() and dist.get_world_size(self.group) > 1:
            self.works.append(dist.all_reduce(self.flat.grad[a:b], op=dist.ReduceOp.SUM, group=self.group, async_op=True))

    # ---- staged form (engine.Trainer: the backward runs as three autograd stages, or three hipGraph replays): bucket k's
    # gradients are complete in the flat buffer when stage k has been enqueued -- no hooks involved
    def begin_staged(self):
        self.enabled = False
        self.works = []

    def launch(self, k):
        """all-reduce bucket k asynchronously: torch.distributed's stream waits for everything enqueued on the current
        stream so far (stage k) and then runs beside stage k + 1"""
        if dist.is_available() and dist.is_initialized() and (dist.get_world_size(self.group) > 1 or self.always_reduce):
            a, b = self.ranges[k]
            self.works.append(dist.all_reduce(self.flat.grad[a:b], op=dist.ReduceOp.SUM, group=self.group, async_op=True))
            self.n_launched += 1

    def finish_staged(self):
        """join every collective (the current stream waits); returns 1/world"""
        for w in self.works:
            w.wait()
        self.works = []
        if dist.is_available() and dist.is_initialized():
            return 1.0 / dist.get_world_size(self.group)
        return 1.0

    def begin(self):
        """call right before backward()"""
        self.pending = [len(m) for m in self.members]
        self.works, self.fired = [], [False] * self.nb
        self.enabled = True

    def finish(self):
        """after backward(): reduce whatever did not fire (unused parameters), join every collective; returns 1/world"""
        self.enabled = False
        for k in range(self.nb):
            if not self.fired[k]:
                self._fire(k)
        for w in self.works:
            w.wait()
        self.works = []
        if dist.is_available() and dist.is_initialized():
            return 1.0 / dist.get_world_size(self.group)
        return 1.0


# readiness order of ModeT's parameters during backward (reference ModeT/models.py:377-412 run in reverse)
MODET_BUCKETS = (
    ("projblock", "mdt", "cwm"),                                     # per-level heads: ready before the encoder's backward starts
    ("encoder.conv4", "encoder.conv3", "encoder.conv2"),             # coarse encoder levels: 86 % of the parameters
    ("encoder.conv1", "encoder.conv0"),                              # full / half resolution blocks: last
)
