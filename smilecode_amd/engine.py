"""Train / inference step of the hot path with the reference's hyper-parameters
(ModeT/train.py:42-168): NCC + Grad3d('l2') with weights [1,1], Adam(lr 1e-4, amsgrad=True),
poly learning-rate schedule, batch of volume pairs per rank, RCCL gradient all-reduce when >1 rank."""
from __future__ import annotations

import contextlib
import os

import torch

from . import ops
from .losses import Grad3d, NCC_vxm
from .parallel import FlatParams, broadcast_parameters


# the step's backward starts from (y_moved, flow) with the gradients the loss kernels wrote (Trainer._seeded_loss); False: from
# the scalar loss through the autograd nodes of ops.ncc_loss / ops.grad3d_loss (read when a Trainer is constructed;
# tools/ab_graphs.py attr:smilecode_amd.engine.SEED_BACKWARD=True,False times the two in one process)
SEED_BACKWARD = True


def poly_lr(epoch, max_epoch=30, init_lr=1e-4, power=0.9):
    """adjust_learning_rate (reference train.py:166-168)"""
    return round(init_lr * (1 - epoch / max_epoch) ** power, 8)


class Trainer:
    def __init__(self, model, lr=1e-4, max_epoch=30, weights=(1.0, 1.0), betas=(0.9, 0.999), eps=1e-8, group=None,
                 overlap_allreduce=False):
        """``overlap_allreduce``: all-reduce the gradients in three buckets while the rest of the backward runs
        (BASELINE.json configs[4]).  The backward is cut into THREE AUTOGRAD STAGES at the bucket boundaries
        (parallel.MODET_BUCKETS: per-level heads | encoder levels 3-5 | encoder levels 1-2; cut tensors = the encoder's
        per-level features and the pooled input of level 3): stage k writes bucket k's gradients into the flat buffer, then
        its ``all_reduce(async_op=True)`` is launched and stage k + 1 runs beside it.  Eager steps run the stages directly;
        ``capture()`` captures them as three hipGraphs sharing one memory pool and a step is replay 0 -> all-reduce 0 ->
        replay 1 -> all-reduce 1 -> replay 2 -> all-reduce 2 -> join -> Adam (host enqueue ~0.3 ms, no Python in the step).
        Without it: one all-reduce of the whole 4 MB buffer after backward / after the single graph's replay."""
        self.model = model
        self.lr0, self.max_epoch, self.weights = lr, max_epoch, weights
        self.betas, self.eps, self.group = betas, eps, group
        self.fp = FlatParams(model)
        broadcast_parameters(self.fp, 0, group)
        self.m = torch.zeros_like(self.fp.flat)
        self.v = torch.zeros_like(self.fp.flat)
        self.vmax = torch.zeros_like(self.fp.flat)
        self.step = 0
        self._graph = self._static_in = self._static_out = self._graph_key = None
        self._stage_graphs = None           # overlap_allreduce: the three captured stage graphs
        self._steps = {}                    # (shape, device, grad mode) -> ops.StepContext, least recently used first
        self.max_step_contexts = 4
        self.batch_small_launches = os.environ.get("MODET_STEP_BATCHING", "1") != "0"
        self.seed_backward = SEED_BACKWARD  # False: the step goes through loss(...)[0].backward() (tests compare the two)
        self.lr_last = lr
        self.sim = NCC_vxm()
        self.reg = Grad3d(penalty="l2")
        self.buckets = None
        if overlap_allreduce:
            from .parallel import MODET_BUCKETS, BucketedAllReduce
            self.buckets = BucketedAllReduce(self.fp, list(model.named_parameters()), MODET_BUCKETS, group)

    def loss(self, moving, fixed):
        y_moved, flow = self.model(moving, fixed)
        sim = self.sim(fixed, y_moved)
        reg = self.reg(flow, fixed)
        if self.weights[0] != 1.0:          # train.py:127-129 multiplies by weights [1, 1]: x * 1.0 == x, and each scalar
            sim = sim * self.weights[0]     # multiplication is a 4 us launch forward and another one backward
        if self.weights[1] != 1.0:
            reg = reg * self.weights[1]
        return sim + reg, sim, reg

    def _seedable(self):
        """the step's own loss path applies: the reference's two loss terms as the HIP kernels have them (cubic NCC window of
        3 / 5 / 7 / 9 voxels, Grad3d without ``loss_mult``) on a model that hands out its channels-last results"""
        return (self.seed_backward and type(self.sim) is NCC_vxm and type(self.reg) is Grad3d and self.reg.loss_mult is None
                and len(set(self.sim._w)) == 1 and self.sim._w[0] in (3, 5, 7, 9) and hasattr(self.model, "forward_cl"))

    def _seeded_loss(self, moving, fixed):
        """``loss`` for the step itself: ((loss, sim, reg) detached, roots, seeds) with ``seeds[i]`` = d loss / d ``roots[i]``
        -- the two loss kernels write value AND weighted gradient in one call (they always did: the autograd nodes of
        ops.ncc_loss / ops.grad3d_loss save the gradient in forward and multiply it by the upstream scalar in backward), so the
        backward starts from y_moved and the flow instead of from the scalar.  Gone from the step: the planar copy of the
        flow (reference losses.py:11-13 indexes (B,3,D,H,W); the kernel reads the channels-last flow the last composition
        wrote) and the copy of its gradient back, the two d * 1.0 passes, autograd's ``ones_like`` root.  Same arithmetic
        element for element: gradients bit-identical to ``loss(...)[0].backward()`` up to the warp scatter's atomic order."""
        y_cl, flow_cl = self.model.forward_cl(moving, fixed)
        B, D, H, W, _ = y_cl.shape
        w0, w1 = float(self.weights[0]), float(self.weights[1])
        sim, d_y = ops.ncc_value_and_grad(fixed.contiguous(), y_cl.detach().reshape(B, 1, D, H, W), self.sim._w[0], w0)
        reg, d_flow = ops.grad3d_value_and_grad_cl(flow_cl.detach(), self.reg.penalty, w1)
        if w0 != 1.0:
            sim = sim * w0
        if w1 != 1.0:
            reg = reg * w1
        return (sim + reg, sim, reg), [y_cl, flow_cl], [d_y.reshape(y_cl.shape), d_flow]

    # ---------------------------------------------------------------- hipGraph replay of forward + backward
    def _fwd_bwd(self, moving, fixed):
        self.fp.zero_grad()
        seeded = self._seedable()
        if not self.batch_small_launches:               # MODET_STEP_BATCHING=0: every conv packs / reduces on its own (A/B)
            with ops.trace_range("forward+loss"):
                if seeded:
                    (loss, sim, reg), roots, seeds = self._seeded_loss(moving, fixed)
                else:
                    loss, sim, reg = self.loss(moving, fixed)
            with ops.trace_range("backward"):
                if seeded:
                    torch.autograd.backward(roots, seeds)
                else:
                    loss.backward()
                self.fp.gather_grads()
            return loss.detach(), sim.detach(), reg.detach()
        # one caller-owned step context per computation this trainer has run (shape, device, grad mode): its recorded
        # packing jobs and its packed-weights arena stay valid for as long as the trainer lives, so a captured graph of
        # shape A keeps working however many other shapes run eagerly in between
        sc = self._step_context(moving)
        # the parameters are constant from here to the end of backward: pack all conv weights in one launch up front
        with sc.prepacked():
            with ops.trace_range("forward+loss"):
                if seeded:
                    (loss, sim, reg), roots, seeds = self._seeded_loss(moving, fixed)
                else:
                    loss, sim, reg = self.loss(moving, fixed)
            with ops.trace_range("backward"):
                # the ~20 per-layer partial-tile reductions as one launch, written straight into the flat gradient buffer
                with sc.deferred(self.fp.grad_destinations()) as scope:
                    if seeded:
                        torch.autograd.backward(roots, seeds)
                    else:
                        loss.backward()
                self.fp.gather_grads(scope.written)
        return loss.detach(), sim.detach(), reg.detach()

    # ---------------------------------------------------------------- backward in three stages (overlapped all-reduce)
    def _step_context(self, moving):
        key = (tuple(moving.shape), moving.device, torch.is_grad_enabled())
        sc = self._steps.pop(key, None)
        if sc is None:
            sc = ops.StepContext()
        self._steps[key] = sc
        while len(self._steps) > self.max_step_contexts:
            for k in self._steps:
                if k != key and not (self._graph_key is not None and k[:2] == self._graph_key):
                    del self._steps[k]
                    break
            else:
                break
        return sc

    def _staged_forward(self, moving, fixed):
        """forward + losses with the autograd graph cut at the bucket boundaries (models.ModeT.stage_cuts: the heads see
        leaves that share the encoder features' storage; Encoder.stage_cut: the same in front of level 3); returns the
        state the three backward stages hand on"""
        m, enc = self.model, self.model.encoder
        m.stage_cuts, enc.stage_cut = True, True
        try:
            if self._seedable():
                (loss, sim, reg), roots, seeds = self._seeded_loss(moving, fixed)
            else:
                loss, sim, reg = self.loss(moving, fixed)
                roots, seeds = [loss], [None]
        finally:
            m.stage_cuts, enc.stage_cut = False, False
        (M, Fx), (Ml, Fl) = m.cut_features, m.cut_leaves
        p3, p3leaf = enc.cut
        m.cut_features = m.cut_leaves = enc.cut = None
        return {"roots": roots, "seeds": seeds, "out": (loss.detach(), sim.detach(), reg.detach()), "feat": list(M) + list(Fx),
                "leaf": list(Ml) + list(Fl), "p3": p3, "p3leaf": p3leaf, "g": None}

    def _staged_backward(self, st, k, sc):
        """stage k of the backward: gradients of bucket k's parameters -> flat buffer (conv weights through the deferred
        reductions of this stage's own scope); gradients of the cut leaves -> st["g"] for the stages that own their producers"""
        members = self.buckets.members[k]
        params = [self.fp.params[i] for i in members]
        if k == 0:                                      # loss -> heads -> feature leaves
            outs, gouts, extra = st["roots"], st["seeds"], st["leaf"]
        else:
            lv = (2, 3, 4) if k == 1 else (0, 1)        # encoder levels 3-5, then 1-2 (feat = [M1..M5, F1..F5])
            idx = [i for i in lv] + [5 + i for i in lv]
            idx = [i for i in idx if st["g"][i] is not None]             # (a feature no loss term reaches has no gradient)
            outs, gouts = [st["feat"][i] for i in idx], [st["g"][i] for i in idx]
            extra = [st["p3leaf"]] if k == 1 else []
            if k == 2 and st["gp3"] is not None:
                outs.append(st["p3"]); gouts.append(st["gp3"])
        with sc.deferred(self.fp.grad_destinations()) as scope:
            grads = torch.autograd.grad(outs, params + extra, gouts, allow_unused=True)
        views, srcs, missing = [], [], []
        for i, gi in zip(members, grads[:len(params)]):
            off, n = self.fp.offsets[i]
            v = self.fp.grad[off:off + n].view(self.fp.params[i].shape)
            if gi is not None:
                if self.fp.params[i].data_ptr() in scope.written:           # a second use of the weight went through autograd
                    v.add_(gi)
                else:
                    views.append(v); srcs.append(gi)
            elif self.fp.params[i].data_ptr() not in scope.written:
                missing.append(v)
        if views:
            torch._foreach_copy_(views, srcs)
        if missing:
            torch._foreach_zero_(missing)
        if k == 0:
            st["g"] = list(grads[len(params):])
            st["roots"] = st["seeds"] = None
        elif k == 1:
            st["gp3"] = grads[len(params)]
            for i in idx:
                st["g"][i] = None                                            # consumed: let the buffers go

    def _fwd_bwd_staged(self, moving, fixed, launch=None):
        """eager form: forward, then stage 0..2, ``launch(k)`` after each (the bucket's all-reduce)"""
        sc = self._step_context(moving)
        with sc.prepacked():
            with ops.trace_range("forward+loss"):
                st = self._staged_forward(moving, fixed)
            for k in range(3):
                with ops.trace_range("backward stage %d" % k):
                    self._staged_backward(st, k, sc)
                if launch is not None:
                    launch(k)
        return st["out"]

    def _capture_staged(self, moving, fixed, warmup, verify):
        """three hipGraphs sharing one pool: [forward + losses + stage 0] [stage 1] [stage 2]"""
        import torch.distributed as dist
        self._static_in = self._static_pair(moving, fixed)
        side = torch.cuda.Stream(device=moving.device)
        side.wait_stream(torch.cuda.current_stream(moving.device))
        with torch.cuda.stream(side):
            for _ in range(max(warmup, 2)):
                self._fwd_bwd_staged(*self._static_in)
        torch.cuda.current_stream(moving.device).wait_stream(side)
        torch.cuda.synchronize(moving.device)
        ref = self.fp.grad.clone() if verify else None
        mode = "thread_local" if (dist.is_available() and dist.is_initialized()) else "global"
        sc = self._step_context(self._static_in[0])
        graphs = [torch.cuda.CUDAGraph() for _ in range(3)]
        # the weight-packing launch is the FIRST NODE of graph 0 (every replay packs the parameters as they are then; packed
        # once outside the graphs, every step after the first Adam update would run its convolutions on the weights of
        # capture time -- ADVICE r4) and the packed-weights scope stays open across the three captures
        with contextlib.ExitStack() as packed:
            with torch.cuda.graph(graphs[0], capture_error_mode=mode):
                packed.enter_context(sc.prepacked())
                st = self._staged_forward(*self._static_in)
                self._staged_backward(st, 0, sc)
            pool = graphs[0].pool()
            for k in (1, 2):
                with torch.cuda.graph(graphs[k], pool=pool, capture_error_mode=mode):
                    self._staged_backward(st, k, sc)
        self._static_out = st["out"]
        st["roots"] = st["seeds"] = None
        self._stage_graphs, self._graph = graphs, graphs[0]
        self._graph_key = (tuple(moving.shape), moving.device)
        if verify:
            self._verify_replay(ref, None, stages=graphs)

    @staticmethod
    def _static_pair(moving, fixed):
        """the captured step's input buffers as the two halves of ONE allocation: the model's [moving; fixed] encoder batch is
        then a view (ops.cat_batch), not a copy"""
        B = moving.shape[0]
        pair = torch.empty((2 * B,) + tuple(moving.shape[1:]), dtype=moving.dtype, device=moving.device)
        pair[:B].copy_(moving)
        pair[B:].copy_(fixed)
        return pair[:B], pair[B:]

    def _verify_tolerance(self):
        """(relative L2 bound per parameter tensor, absolute floor as a fraction of the largest tensor's norm).  A healthy fp32
        replay differs from the eager step by the float-atomic reorder of the warp scatter, ~1e-6; with bf16 storage a flipped
        rounding downstream makes it a few % of a tiny-gradient tensor's norm.  A broken replay (stale buffer, a node that
        did not replay as it ran) is anything above that: round 4 accepted 25 % in every mode and let a defect through."""
        if getattr(self.model, "act_dtype", torch.float32) == torch.bfloat16:
            return 0.25, 1e-4
        return 1e-4, 1e-5

    def _verify_replay(self, ref, replay, stages=None):
        """``replay()`` twice against the eager gradients ``ref``, PER PARAMETER TENSOR (relative L2: a replay that is wrong only in
        a tensor whose gradients are 100x below the global maximum must not pass on the strength of the large ones).
        ``stages``: the list of stage graphs of the overlapped form -- each is replayed on its own and must (a) produce its
        bucket's gradients and (b) leave the buckets of the earlier stages BIT FOR BIT as they were: their all-reduce is in
        flight while this stage runs."""
        seg = self.fp.segment_index()
        nseg = len(self.fp.params)
        rel, floor_frac = self._verify_tolerance()
        ref_sq = torch.zeros(nseg, device=ref.device, dtype=torch.float64).index_add_(0, seg, ref.double() ** 2)
        # analytically-zero gradients (a conv bias under InstanceNorm) are pure rounding noise: they are held to an absolute
        # level relative to the largest tensor's norm instead
        floor = floor_frac ** 2 * float(ref_sq.max())
        names = [n for n, p in self.model.named_parameters() if p.requires_grad]

        def fail(rep, what):
            self.release_graph()
            raise RuntimeError(f"hipGraph replay {rep} of the train step does not reproduce the eager gradients: {what}; "
                               f"running eagerly is the fallback")

        for rep in range(2):
            self.fp.grad.fill_(float("nan"))
            if stages is None:
                replay()
            else:
                done = []
                for k, gr in enumerate(stages):
                    gr.replay()
                    for j, (a, b), snap in done:
                        if not torch.equal(self.fp.grad[a:b], snap):
                            fail(rep, f"stage {k} modified bucket {j}, whose all-reduce is in flight by then")
                    a, b = self.buckets.ranges[k]
                    if not bool(torch.isfinite(self.fp.grad[a:b]).all()):
                        fail(rep, f"stage {k} left bucket {k} incomplete")
                    done.append((k, (a, b), self.fp.grad[a:b].clone()))
            d = (self.fp.grad - ref).double() ** 2
            err_sq = torch.zeros(nseg, device=ref.device, dtype=torch.float64).index_add_(0, seg, d)   # NaN propagates
            bad = torch.nonzero(~(err_sq <= rel ** 2 * ref_sq + floor)).flatten().tolist()
            if bad:
                i = bad[0]
                where = "" if self.buckets is None else f" [stage {self.buckets.bucket_of[i]}]"
                fail(rep, f"{len(bad)} of {nseg} parameter tensors differ, first #{i} {names[i]}{where} (|diff|_2 "
                          f"{float(err_sq[i]) ** 0.5:.3e} vs |grad|_2 {float(ref_sq[i]) ** 0.5:.3e}, bound {rel:g})")

    def capture(self, moving, fixed, warmup=2, verify=True):
        """Capture forward + losses + backward + gradient packing for this input shape into ONE hipGraph
        (torch.cuda.CUDAGraph = hipGraph on ROCm).  A step is ~500 kernel launches issued from Python through ctypes and
        the autograd tape (6-7 ms of host time at 160x192x160); replaying the graph costs the host ~0.1 ms, so the GPU
        never waits for Python however fast the kernels get.  The gradient all-reduce and the fused Adam kernel stay
        outside the graph (their lr / step / 1/world arguments change per step; RCCL picks its own stream order).
        ``warmup`` eager steps of the SAME computation run first on the capture stream (every kernel must have been
        launched once: lazy code-object loading and the occupancy memo are not capturable); they do not update the
        parameters.  ``verify``: the captured step is replayed TWICE and its gradients compared with the eager step's
        before the graph is accepted (RuntimeError otherwise) -- a captured stream operation that is not a kernel may replay
        differently from how it ran (round 3: a hipMemsetAsync node cleared its buffer on the first replay only and the step
        was silently wrong from the second one on); two eager-sized steps, once per capture.  Returns self."""
        self.model.train()
        if self.buckets is not None:
            self._capture_staged(moving, fixed, warmup, verify)
            return self
        self._static_in = self._static_pair(moving, fixed)
        side = torch.cuda.Stream(device=moving.device)
        side.wait_stream(torch.cuda.current_stream(moving.device))
        with torch.cuda.stream(side):
            # with step batching the first pass of a shape only RECORDS the packing jobs: a second pass must launch the
            # batched packing kernels (and read the arena) once before they can be captured
            for _ in range(max(warmup, 2 if self.batch_small_launches else 1)):
                self._fwd_bwd(*self._static_in)
        torch.cuda.current_stream(moving.device).wait_stream(side)
        torch.cuda.synchronize(moving.device)
        self.fp.zero_grad()
        g = torch.cuda.CUDAGraph()
        # with a process group alive, RCCL's watchdog thread polls events while we capture: only THIS thread's calls may
        # invalidate the capture then (kernels the autograd thread launches into the capturing stream are captured either way)
        import torch.distributed as dist
        mode = "thread_local" if (dist.is_available() and dist.is_initialized()) else "global"
        with torch.cuda.graph(g, capture_error_mode=mode):
            self._static_out = self._fwd_bwd(*self._static_in)
        self._graph = g
        self._graph_key = (tuple(moving.shape), moving.device)
        if verify:
            ref = self.fp.grad.clone()                      # the last warm-up pass = the eager step on these parameters
            self._verify_replay(ref, g.replay)
        return self

    def release_graph(self):
        self._graph = self._static_in = self._static_out = self._graph_key = self._stage_graphs = None

    def release_steps(self):
        """drop every cached step context (recorded packing jobs + packed-weights arenas) that no captured graph uses"""
        keep = {k: v for k, v in self._steps.items() if self._graph_key is not None and k[:2] == self._graph_key}
        self._steps = keep

    def train_step(self, moving, fixed, epoch=0):
        """one iteration of train.py:114-133; returns device scalars (no host sync)"""
        self.model.train()
        if getattr(self, "_graph", None) is not None and self._graph_key == (tuple(moving.shape), moving.device):
            self._static_in[0].copy_(moving, non_blocking=True)
            self._static_in[1].copy_(fixed, non_blocking=True)
            if self._stage_graphs is not None:              # overlapped: replay k -> all-reduce of bucket k (async) -> replay k + 1
                self.buckets.begin_staged()
                for k, gr in enumerate(self._stage_graphs):
                    gr.replay()
                    self.buckets.launch(k)
                scale = self.buckets.finish_staged()
            else:
                self._graph.replay()
                scale = self.fp.allreduce_grads(self.group)
            loss, sim, reg = self._static_out
        elif self.buckets is not None:
            self.buckets.begin_staged()
            loss, sim, reg = self._fwd_bwd_staged(moving, fixed, self.buckets.launch)
            scale = self.buckets.finish_staged()
        else:
            loss, sim, reg = self._fwd_bwd(moving, fixed)
            scale = self.fp.allreduce_grads(self.group)
        self.step += 1
        self.lr_last = poly_lr(epoch, self.max_epoch, self.lr0)
        ops.adam_amsgrad_step_(self.fp.flat, self.fp.grad, self.m, self.v, self.vmax, self.lr_last, self.step,
                               self.betas[0], self.betas[1], self.eps, scale)
        return loss, sim, reg

    # ---------------------------------------------------------------- optimizer state, torch.optim.Adam's format
    def state_dict(self):
        """The dict ``torch.optim.Adam(model.parameters(), amsgrad=True).state_dict()`` would hold at this point -- what the
        reference saves under ``'optimizer'`` (train.py:158-163): ``state[i] = {step, exp_avg, exp_avg_sq,
        max_exp_avg_sq}`` per parameter (in ``model.parameters()`` order, sliced out of the flat fused buffers) and one
        ``param_groups`` entry.  A consumer can ``optimizer.load_state_dict()`` it into a real torch Adam."""
        state = {}
        if self.step > 0:
            for i, (p, (off, k)) in enumerate(zip(self.fp.params, self.fp.offsets)):
                state[i] = {"step": torch.tensor(float(self.step)),
                            "exp_avg": self.m[off:off + k].view(p.shape).clone(),
                            "exp_avg_sq": self.v[off:off + k].view(p.shape).clone(),
                            "max_exp_avg_sq": self.vmax[off:off + k].view(p.shape).clone()}
        group = {"lr": self.lr_last, "betas": tuple(self.betas), "eps": self.eps, "weight_decay": 0, "amsgrad": True,
                 "maximize": False, "foreach": None, "capturable": False, "differentiable": False, "fused": None,
                 "params": list(range(len(self.fp.params)))}
        return {"state": state, "param_groups": [group]}

    def load_state_dict(self, sd):
        """restore Adam's moments and step count from ``state_dict()``'s format (ours or a checkpoint the reference's
        torch.optim.Adam wrote for the same model)"""
        if "state" not in sd or "param_groups" not in sd:
            raise RuntimeError("Trainer.load_state_dict: not a torch.optim.Adam state dict (keys: %s)" % sorted(sd))
        st = sd["state"]
        n = len(self.fp.params)
        if len(st) not in (0, n):
            raise RuntimeError(f"Trainer.load_state_dict: optimizer state has {len(st)} entries, the model {n} parameters")
        self.m.zero_(); self.v.zero_(); self.vmax.zero_()
        self.step = 0
        steps = set()
        for i, (p, (off, k)) in enumerate(zip(self.fp.params, self.fp.offsets)):
            e = st.get(i, st.get(str(i)))
            if e is None:
                continue
            if e.get("max_exp_avg_sq", None) is None:
                raise RuntimeError("Trainer.load_state_dict: the checkpoint's Adam was not amsgrad=True")
            for buf, key in ((self.m, "exp_avg"), (self.v, "exp_avg_sq"), (self.vmax, "max_exp_avg_sq")):
                t = e[key]
                if tuple(t.shape) != tuple(p.shape):
                    raise RuntimeError(f"Trainer.load_state_dict: parameter {i}: {key} {tuple(t.shape)} vs {tuple(p.shape)}")
                buf[off:off + k].copy_(t.reshape(-1).to(buf.device, torch.float32))
            steps.add(int(float(e["step"])))
        if len(steps) > 1:
            raise RuntimeError(f"Trainer.load_state_dict: parameters disagree on the step count: {sorted(steps)}")
        if steps:
            self.step = steps.pop()
        g = sd["param_groups"][0]
        self.betas, self.eps = tuple(float(v) for v in g.get("betas", self.betas)), float(g.get("eps", self.eps))
        self.lr_last = float(g.get("lr", self.lr_last))     # the reference stores a numpy.float64 here (train.py:166-168)

    @torch.no_grad()
    def infer(self, moving, fixed):
        self.model.eval()
        return self.model(moving, fixed)
