"""Torch-facing operators of the ModeT hot path: thin autograd wrappers over libmodet_hip.so.

PyTorch is plumbing here (device memory from the caching allocator, the current HIP
stream, autograd's tape); every arithmetic kernel is hand-written HIP behind the C ABI of
include/modet_hip.h.  Activations are channels-last ``(B, D, H, W, C)`` fp32 (bf16 inside the ConvInsBlock chains when the model is built with
``act_dtype=torch.bfloat16``, see the end of this file), contiguous.
There is no eager / CPU fallback: a missing library or a non-GPU tensor raises.
"""
from __future__ import annotations

import ctypes
import os
import threading

import torch
from torch.autograd import Function

from . import _lib

LRELU_SLOPE = 0.1


def _L():
    return _lib.load()


def _stream():
    return torch.cuda.current_stream().cuda_stream


def _chk(*ts):
    for t in ts:
        if t is None:
            continue
        if not t.is_cuda:
            raise RuntimeError("smilecode_amd: tensor must live on the GPU (the HIP path has no CPU fallback)")
        if t.dtype != torch.float32:
            raise RuntimeError(f"smilecode_amd: expected float32, got {t.dtype}")
        if not t.is_contiguous():
            raise RuntimeError("smilecode_amd: tensor must be contiguous")


def _p(t):
    return None if t is None else t.data_ptr()


def _ws(nbytes, like):
    return torch.empty((int(nbytes) + 3) // 4 + 1, dtype=torch.float32, device=like.device)


class KernelTimer:
    """Optional per-launch timing with HIP events on the launch stream (bench.py's live roofline leg).
    ``select`` = None times every tagged launch, or a set of tags to time only those."""

    def __init__(self, select=None):
        self.select = select
        self.records = []          # (tag, flops, bytes, start_event, end_event)

    def summary(self):
        """tag -> dict(calls, ms, flops, bytes); synchronises the events it reads"""
        out = {}
        for tag, fl, by, e0, e1 in self.records:
            e1.synchronize()
            d = out.setdefault(tag, {"calls": 0, "ms": 0.0, "flops": 0.0, "bytes": 0.0})
            d["calls"] += 1
            d["ms"] += e0.elapsed_time(e1)
            d["flops"] += fl
            d["bytes"] += by
        return out


_TIMER = None


def set_kernel_timer(timer):
    """install / remove (None) the KernelTimer consulted by every launch below"""
    global _TIMER
    _TIMER = timer


class _Guard:
    """make x's device current for the launch (autograd worker threads, multi-GPU hosts); when a
    KernelTimer is installed, bracket the launch with HIP events on the current stream."""

    def __init__(self, t, tag=None, flops=0.0, nbytes=0.0):
        self.idx = t.device.index
        self.prev = None
        self.tag, self.flops, self.nbytes = tag, flops, nbytes
        self.e0 = None

    def __enter__(self):
        cur = torch.cuda.current_device()
        if cur != self.idx:
            self.prev = cur
            torch.cuda.set_device(self.idx)
        tm = _TIMER
        if tm is not None and self.tag is not None and (tm.select is None or self.tag in tm.select):
            self.e0 = torch.cuda.Event(enable_timing=True)
            self.e0.record()

    def __exit__(self, *a):
        if self.e0 is not None:
            e1 = torch.cuda.Event(enable_timing=True)
            e1.record()
            _TIMER.records.append((self.tag, self.flops, self.nbytes, self.e0, e1))
        if self.prev is not None:
            torch.cuda.set_device(self.prev)


class _Roctx:
    """roctx ranges around the pyramid levels / phases of a step (SURVEY.md §5 tracing row): visible in
    ``rocprofv3 --marker-trace`` timelines.  Off unless MODET_ROCTX=1 (two ctypes calls per range otherwise wasted)."""

    def __init__(self):
        import ctypes
        import os
        self.lib = None
        if os.environ.get("MODET_ROCTX") == "1":
            for name in ("libroctx64.so", "/opt/rocm/lib/libroctx64.so", "librocprofiler-sdk-roctx.so"):
                try:
                    self.lib = ctypes.CDLL(name)
                    self.lib.roctxRangePushA.argtypes = [ctypes.c_char_p]
                    break
                except OSError:
                    continue

    def push(self, name):
        if self.lib is not None:
            self.lib.roctxRangePushA(name.encode())

    def pop(self):
        if self.lib is not None:
            self.lib.roctxRangePop()


_ROCTX = None


class trace_range:
    """``with ops.trace_range("level3"):`` -- a roctx range when MODET_ROCTX=1, nothing otherwise"""

    def __init__(self, name):
        self.name = name

    def __enter__(self):
        global _ROCTX
        if _ROCTX is None:
            _ROCTX = _Roctx()
        _ROCTX.push(self.name)

    def __exit__(self, *a):
        _ROCTX.pop()


_FAM_CACHE = {}
_FAM_SUFFIX = ("", "@split", "@x3", "@direct", "@tr", "@q")


def _conv_tag(kind, x_shape, Cin, Cout, variant=0, f16=None):
    """KernelTimer tag of a conv launch: 'conv_fwd[8->8]@x3' names the kernel family that runs this shape (exact-f32
    MFMA: no suffix; tiled bf16x3: @split; z-marching bf16x3: @x3; small-volume direct MFMA: @direct; transpose-read weight
    gradient: @tr) -- only evaluated while a timer is installed.  ``variant`` = what the launch fuses (it changes the routing,
    include/modet_hip.h modet_conv3d_kernel_family_v): 0 plain, 1 LeakyReLU, 2 normalised input, 3 statistics.  '@x3h': the
    z-marching kernel on two f16 pieces (three products per fp32 product instead of six) -- every forward launch, and the
    backward launches that were given max |d_y| (f16=True)."""
    if _TIMER is None:
        return None
    f16 = (kind == "fwd") if f16 is None else bool(f16)
    key = (kind, tuple(x_shape[:4]), Cin, Cout, variant, f16)
    t = _FAM_CACHE.get(key)
    if t is None:
        B, D, H, W = x_shape[:4]
        fam = _L().modet_conv3d_kernel_family_v(B, D, H, W, Cin, Cout, {"fwd": 0, "dgrad": 1, "wgrad": 2}[kind], variant)
        arrow = f"{Cout}->{Cin}" if kind == "dgrad" else f"{Cin}->{Cout}"
        half = f16 and D * H * W < (1 << 24) and fam in (2, 4, 5)
        t = _FAM_CACHE[key] = f"conv_{kind}[{arrow}]{_FAM_SUFFIX[fam]}" + ("h" if half else "")
    return t


def _conv16_tag(kind, x_shape, Cin, Cout, x_bf16):
    """the same for the bf16-storage entry points: 'conv_bf16_fwd[8->8]@x3' = the one-piece z-marching kernel"""
    if _TIMER is None:
        return None
    key = ("b16" + kind, tuple(x_shape[:4]), Cin, Cout, bool(x_bf16))
    t = _FAM_CACHE.get(key)
    if t is None:
        B, D, H, W = x_shape[:4]
        fam = _L().modet_conv3d_bf16_kernel_family(B, D, H, W, Cin, Cout, {"fwd": 0, "dgrad": 1, "wgrad": 2}[kind], int(x_bf16))
        arrow = f"{Cout}->{Cin}" if kind == "dgrad" else f"{Cin}->{Cout}"
        t = _FAM_CACHE[key] = f"conv_bf16_{kind}[{arrow}]" + ("@x3" if fam == 2 else "")
    return t


# ------------------------------------------------------------------------------------------------ raw calls
# The one un-normalised activation of the model is the ConvBlock 1 -> 4 output in front of the first ConvInsBlock (reference
# models.py:192).  Default (False): that 4 -> 8 layer runs the three bf16 pieces -- range-free, and the most accurate form: flow
# error vs fp64 at 160x192x160 4.3e-4 voxels, worst gradient 6.4e-4 of its tensor's max.  True: the first block's kernel leaves
# max |y| on the device and the 4 -> 8 layer (forward + weight gradient) scales two f16 pieces by it -- also range-free, 0.05 ms
# per step faster (6.94 vs 6.99), but 7.2e-4 / 4.4e-3 (measured, profiles/r06*_first_block_f16.txt): not worth the margin.
FIRST_BLOCK_F16 = False


def conv3d_forward(x, w, b, act, step=None, x_act=False):
    """x_act: the caller's word that x is an activation inside the f16 forms' range (include/modet_hip.h, "TWO f16 PIECES":
    |x| < 4 094) -> the *_bounded entry point, half the matrix work; without it the launch makes no assumption about x (three
    bf16 pieces: fp32's range), as nn.Conv3d makes none"""
    _chk(x, w, b)
    step = step if step is not None else current_step()
    B, D, H, W, Cin = x.shape
    Cout = w.shape[0]
    if tuple(w.shape) != (Cout, Cin, 3, 3, 3):
        raise RuntimeError(f"conv3d: weight {tuple(w.shape)} does not match input channels {Cin}")
    y = torch.empty((B, D, H, W, Cout), dtype=torch.float32, device=x.device)
    L = _L()
    nb = L.modet_conv3d_ws_bytes(Cin, Cout)
    ws = _ws(nb, x)
    n = float(B) * D * H * W
    with _Guard(x, _conv_tag("fwd", x.shape, Cin, Cout, 1 if act else 0, f16=bool(x_act)), 54.0 * Cin * Cout * n, 4.0 * n * (Cin + Cout)):
        if FIRST_BLOCK_F16 and Cin == 1 and Cout == 4 and B * D * H * W >= 500000 and not x_act:
            # the first encoder block: its kernel leaves max |y| on the device for free, and the next layer (whose input this
            # un-normalised tensor is) scales its f16 pieces by it -- any image range at the f16 forms' speed
            amax = torch.empty(AMAX_FLOATS, dtype=torch.float32, device=x.device)
            rc = L.modet_conv3d_fwd_amax_out(_p(x), _p(w), _p(b), _p(y), _p(ws), nb, B, D, H, W, Cin, Cout, int(act), _p(amax),
                                             _stream(), _h(step))
            if rc == 0:
                return _tag_xamax(y, amax)
        fn = L.modet_conv3d_fwd_bounded if x_act else L.modet_conv3d_fwd
        _lib.check(fn(_p(x), _p(w), _p(b), _p(y), _p(ws), nb, B, D, H, W, Cin, Cout, int(act), _stream(), _h(step)), "modet_conv3d_fwd")
    return y


def instnorm_stats(x_raw, stats=None, eps=1e-5):
    """per-(sample, channel) mean / rstd of a raw conv output, without the apply pass: from the conv epilogue's partial
    statistics when given, else by one statistics pass over x_raw"""
    _chk(x_raw)
    B, C = x_raw.shape[0], x_raw.shape[-1]
    V = x_raw.numel() // (B * C)
    mean = torch.empty(B * C, dtype=torch.float32, device=x_raw.device)
    rstd = torch.empty_like(mean)
    L = _L()
    with _Guard(x_raw, "instnorm_stats", 2.0 * x_raw.numel(), 0.0 if stats is not None else 4.0 * x_raw.numel()):
        if stats is not None:
            _lib.check(L.modet_instnorm_stats(None, _p(mean), _p(rstd), _p(stats), stats.numel() * 4, None, 0, B, V, C, eps,
                                              _stream()), "modet_instnorm_stats")
        else:
            nb = L.modet_instnorm_ws_bytes(B, V, C)
            ws = _ws(nb, x_raw)
            _lib.check(L.modet_instnorm_stats(_p(x_raw), _p(mean), _p(rstd), None, 0, _p(ws), nb, B, V, C, eps, _stream()),
                       "modet_instnorm_stats")
    return mean, rstd


def conv3d_forward_normin(x_raw, mean, rstd, w, b, want_stats=True, step=None):
    """conv3d(LeakyReLU((x_raw - mean) * rstd), w, b) with the normalisation applied while the input tile is staged;
    returns (y, stats or None)"""
    _chk(x_raw, w, b)
    step = step if step is not None else current_step()
    B, D, H, W, Cin = x_raw.shape
    Cout = w.shape[0]
    L = _L()
    y = torch.empty((B, D, H, W, Cout), dtype=torch.float32, device=x_raw.device)
    nb = L.modet_conv3d_ws_bytes(Cin, Cout)
    ws = _ws(nb, x_raw)
    sb = L.modet_conv3d_normin_stats_bytes(B, D, H, W, Cin, Cout) if want_stats else 0
    stats = torch.empty(sb // 4, dtype=torch.float32, device=x_raw.device) if sb > 0 else None
    n = float(B) * D * H * W
    with _Guard(x_raw, _conv_tag("fwd", x_raw.shape, Cin, Cout, 2), 54.0 * Cin * Cout * n, 4.0 * n * (Cin + Cout)):
        _lib.check(L.modet_conv3d_fwd_normin(_p(x_raw), _p(mean), _p(rstd), _p(w), _p(b), _p(y), _p(ws), nb, _p(stats), sb,
                                             B, D, H, W, Cin, Cout, _stream(), _h(step)), "modet_conv3d_fwd_normin")
    return y, stats


# ---- gradient magnitudes for the f16 forms of the backward convolutions (round 5).  The z-marching kernels run on two f16 pieces
# per operand (three MFMA products instead of the six of three bf16 pieces) when they know max |d_y|: f16 has the mantissa for
# it, not the range.  The InstanceNorm backward that PRODUCES a d_y leaves that maximum in a one-float tensor for free
# (modet_instnorm_lrelu_bwd*_amax) and tags its output with it; the conv backward that CONSUMES the tensor reads the tag.  The tag
# carries the tensor's version counter: if autograd accumulated another gradient into the tensor on the way (in place), the
# maximum no longer bounds it and the consumer falls back to the bf16 pieces.  MODET_GRAD_F16=0: never tag (A/B switch).
GRAD_F16 = os.environ.get("MODET_GRAD_F16", "1") != "0"


AMAX_FLOATS = 64 * 32                     # MODET_AMAX_FLOATS: 64 slots, 128 bytes apart (include/modet_hip.h)


def _new_amax(like):
    """the buffer an InstanceNorm backward leaves max |d_x| in -- only where a conv kernel with an f16 form (families 2 and 5)
    would consume d_x as its d_y, else None (the plain call)"""
    if not GRAD_F16:
        return None
    if like.dim() != 5:
        return None
    B, D, H, W, C = like.shape
    key = ("amax", B, D, H, W, C)
    use = _FAM_CACHE.get(key)
    if use is None:          # (the consumer's other channel count is not known here; the families split by volume and channel class)
        use = _FAM_CACHE[key] = _L().modet_conv3d_kernel_family(B, D, H, W, C, C, 1) in (2, 5)
    return torch.empty(AMAX_FLOATS, dtype=torch.float32, device=like.device) if use else None


def amax_buffer(value):
    """a gradient-maximum buffer (every slot = value) for callers that know a bound of |d_y| themselves"""
    return value.detach().reshape(1).float().expand(AMAX_FLOATS).contiguous()


def _tag_amax(t, amax):
    if amax is not None:
        t._modet_amax = (amax, t._version)
    return t


def _tag_xamax(t, amax):
    """mark an ACTIVATION tensor with the device-side maxima of its magnitude (left by the kernel that produced it)"""
    t._modet_xamax = (amax, t._version)
    return t


def _xamax_of(t):
    tag = getattr(t, "_modet_xamax", None)
    return tag[0] if tag is not None and tag[1] == t._version else None


def _amax_of(t):
    tag = getattr(t, "_modet_amax", None)
    if tag is None or not GRAD_F16 or tag[1] != t._version:
        return None
    return tag[0]


def conv3d_backward_data(dy, w, Cin, step=None, amax=None):
    _chk(dy, w)
    step = step if step is not None else current_step()
    B, D, H, W, Cout = dy.shape
    dx = torch.empty((B, D, H, W, Cin), dtype=torch.float32, device=dy.device)
    L = _L()
    nb = L.modet_conv3d_ws_bytes(Cin, Cout)
    ws = _ws(nb, dy)
    n = float(B) * D * H * W
    with _Guard(dy, _conv_tag("dgrad", dy.shape, Cin, Cout, f16=amax is not None), 54.0 * Cin * Cout * n, 4.0 * n * (Cin + Cout)):
        _lib.check(L.modet_conv3d_bwd_data_amax(_p(dy), _p(w), _p(dx), _p(ws), nb, B, D, H, W, Cin, Cout, _p(amax), _stream(),
                                                _h(step)), "modet_conv3d_bwd_data")
    return dx


class StepContext:
    """Caller-owned state of the step-level launch batching (include/modet_hip.h, modet_step_ctx_t): the recorded
    weight-packing jobs of ONE forward+backward computation (one input shape / grad mode) and the weight-gradient
    reductions its backward pass queues.  Nothing is process-wide: every Trainer keeps its own contexts (one per shape it
    has seen), so two trainers in one process -- two threads, two devices, a training and an evaluation model -- never see
    each other's jobs.

      with sc.prepacked():                       # first use records the packing jobs while running normally; later uses
          forward                                # start with ONE launch that packs every recorded weight tensor and the
          with sc.deferred(dst) as scope:        # conv launches skip theirs
              backward                           # ~20 partial-tile reductions -> one launch at scope exit, written to dst

    The scopes bind the context to the calling thread; every conv op picks it up in its forward and hands it to its own
    backward (which runs on the autograd thread), so no op ever consults a global.  The owner guarantees that the weights
    are not modified inside ``prepacked()`` and live at the same addresses from one pass to the next (FlatParams views
    do).  The packed-weights arena lives as long as the context -- a captured hipGraph that bakes its address in stays
    valid however many other shapes the trainer runs eagerly in between."""

    def __init__(self):
        import ctypes
        h = ctypes.c_void_p()
        _lib.check(_L().modet_step_ctx_create(ctypes.byref(h)), "modet_step_ctx_create")
        self.handle = h
        self.arena = None            # packed weights of every recorded conv launch (fp32 jobs, then the 16-bit ones)
        self.recorded = False
        self.dst = None              # parameter.data_ptr() -> gradient destination, while a deferred() scope is open
        self.written = set()
        self._keep = []
        self._leaf = []              # queued leaf reductions (modet_leaf_job_t) of the open deferred() scope, see defer_leaf
        self._side = {}              # device index -> the stream the small levels' weight gradients run on (see side_stream)
        self._side_used = None

    def __del__(self):
        h, self.handle = getattr(self, "handle", None), None
        lib = getattr(_lib, "_lib", None) if _lib is not None else None     # (module globals are gone at interpreter exit)
        if h is not None and lib is not None:
            lib.modet_step_ctx_destroy(h)

    class _Bind:
        def __init__(self, sc):
            self.sc = sc

        def __enter__(self):
            self.prev = getattr(_TLS, "step", None)
            _TLS.step = self.sc
            return self

        def __exit__(self, *exc):
            _TLS.step = self.prev
            return False

    class _Prepacked(_Bind):
        def __enter__(self):
            sc, L = self.sc, _L()
            self.recording = not sc.recorded
            if self.recording:
                L.modet_conv3d_prepack_record(sc.handle, 1)
            else:
                _lib.check(L.modet_conv3d_prepack_begin(sc.handle, _p(sc.arena), sc.arena.numel() * 4, _stream()),
                           "modet_conv3d_prepack_begin")
            return super().__enter__()

        def __exit__(self, *exc):
            sc, L = self.sc, _L()
            super().__exit__(*exc)
            if self.recording:
                L.modet_conv3d_prepack_record(sc.handle, 0)
                if exc[0] is None:
                    dev = torch.device("cuda", torch.cuda.current_device())
                    sc.arena = torch.empty(L.modet_conv3d_prepack_arena_bytes(sc.handle) // 4 + 64, dtype=torch.float32,
                                           device=dev)
                    sc.recorded = True
            else:
                L.modet_conv3d_prepack_end(sc.handle)
            return False

    class _Deferred(_Bind):
        def __init__(self, sc, dst):
            super().__init__(sc)
            self.dst = dst
            self.written = set()

        def __enter__(self):
            sc = self.sc
            if sc.dst is not None:
                raise RuntimeError("StepContext.deferred scopes do not nest")
            sc.dst, sc.written, sc._keep, sc._leaf = self.dst, self.written, [], []
            return super().__enter__()

        def __exit__(self, *exc):
            sc = self.sc
            super().__exit__(*exc)
            sc.dst = None
            if sc._side_used is not None:                # the partial tiles the side stream produced: join before reducing them
                torch.cuda.current_stream().wait_stream(sc._side_used)
                sc._side_used = None
            rc = _L().modet_conv3d_wgrad_defer_flush(sc.handle, _stream())    # always empties the queue, also on an exception
            jobs, sc._leaf = sc._leaf, []
            if exc[0] is None and jobs:                  # the attention / projection parameter gradients of every level: one launch
                arr = (_lib.LeafJob * len(jobs))(*jobs)            # host table: copied into the launch's arguments
                _lib.check(_L().modet_leaf_reduce_many(ctypes.addressof(arr), len(jobs), _stream()), "modet_leaf_reduce_many")
            sc._keep = []
            if exc[0] is None:
                _lib.check(rc, "modet_conv3d_wgrad_defer_flush")
            return False

    def prepacked(self):
        return StepContext._Prepacked(self)

    def deferred(self, dst):
        """``dst`` maps ``parameter.data_ptr()`` to the tensor that parameter's gradient must be written to (FlatParams: a
        view of the flat gradient buffer).  A conv whose weight (and bias) have a destination writes there -- after the
        flush -- and returns no gradient to autograd, so nothing depends on how autograd hands tensors to ``.grad``; the
        pointers written are in ``scope.written``.  A conv without a destination, or a second use of the same weight inside
        one scope, takes the immediate path."""
        return StepContext._Deferred(self, dst)

    def side_stream(self, like):
        """Fork: the stream the weight gradients of the SMALL levels run on, made to wait for everything enqueued on the
        current stream so far (their operands).  Pyramid levels 3-5 and the CWM layers are latency-bound launches of 100-1000
        workgroups on 256 CUs; the data gradient is what the next layer waits for, the weight gradient is needed only at the end
        of the pass (the deferred reduction), so the two chains run side by side -- as two branches of the captured hipGraph.
        The scope's flush joins.  Callers keep the operands alive until then (``_keep``): a tensor freed on the main stream
        could otherwise be handed out again while the side stream still reads it."""
        dev = like.device.index if like.device.index is not None else torch.cuda.current_device()
        side = self._side.get(dev)
        if side is None:
            side = self._side[dev] = torch.cuda.Stream(device=dev)
        side.wait_stream(torch.cuda.current_stream(dev))
        self._side_used = side
        return side

    def destinations(self, w, b, want_bias):
        """(d_w, d_bias) destinations for this call, or None -> immediate path"""
        if self.dst is None:
            return None
        dw = self.dst.get(w.data_ptr()) if w is not None else None
        if dw is None or dw.shape != w.shape or w.data_ptr() in self.written:
            return None
        db = None
        if want_bias:
            db = self.dst.get(b.data_ptr()) if b is not None else None
            if db is None or db.shape != b.shape or b.data_ptr() in self.written:
                return None
        return dw, db

    def defer_leaf(self, ws, outer, outer_stride, rows, row_stride, ncols, col_group, col_group_stride, dsts):
        """queue one column-sum job over the partial rows a backward kernel left at the start of ``ws`` (kept alive until the
        scope's flush): see modet_leaf_reduce_many.  ``dsts``: up to four destination tensors, consecutive column segments."""
        j = _lib.LeafJob()
        j.part, j.outer, j.outer_stride, j.rows, j.row_stride = ws.data_ptr(), outer, outer_stride, rows, row_stride
        j.col_group_stride, j.ncols, j.col_group = col_group_stride, ncols, col_group
        for u in range(4):
            j.dst[u] = dsts[u].data_ptr() if u < len(dsts) else None
            j.n[u] = dsts[u].numel() if u < len(dsts) else 0
        if sum(j.n) != ncols:
            raise RuntimeError("defer_leaf: the destination segments do not add up to the job's columns")
        self._leaf.append(j)
        self._keep.append(ws)

    def claim(self, *params):
        """gradient destinations of a node's parameters (projection weight / bias / gamma / beta, an attention's rpb): the
        list of destination tensors when EVERY one has a destination and none was written in this scope yet -- they are then
        marked written and the node's backward kernel writes its results straight there (no gradient returned to autograd,
        no packing copy) -- else None -> the node returns its gradients to autograd as usual."""
        if self.dst is None:
            return None
        out = []
        for p in params:
            d = self.dst.get(p.data_ptr())
            if d is None or d.shape != p.shape or p.data_ptr() in self.written:
                return None
            out.append(d)
        for p in params:
            self.written.add(p.data_ptr())
        return out


# the parameter gradients of the attention / projection nodes (d_rpb; d_gamma, d_beta, d_bias, d_W) of all levels are summed
# by ONE launch at the end of the deferred() scope instead of 2 + 1 launches per level (read at backward time)
DEFER_LEAF_REDUCTIONS = True

_TLS = threading.local()


def _defer_rpb(step, ws, B, D, H, W, heads, hd, drpb):
    """the d_rpb column sums of one attention backward (partial rows at the start of ``ws``: [B][heads][rows][27]) -> the scope's
    single leaf-reduction launch"""
    rows = int(_L().modet_na_bwd_partial_rows(B, D, H, W, heads, hd))
    step.defer_leaf(ws, B, heads * rows * 27, rows, 27, heads * 27, 27, rows * 27, [drpb])


def _defer_proj(step, ws, N, Cin, dim, dW, db, dg, dbeta):
    """the parameter-gradient column sums of one paired projection backward (rows [d_gamma | d_beta | d_bias | d_W])"""
    rows = int(_L().modet_proj_ln_bwd_pair_partial_rows(N, Cin, dim))
    ncols = 3 * dim + dim * Cin
    step.defer_leaf(ws, 1, 0, rows, ncols, ncols, ncols, 0, [dg, dbeta, db, dW])


def current_step():
    """the StepContext bound to this thread by an open ``prepacked()`` / ``deferred()`` scope, or None"""
    return getattr(_TLS, "step", None)


def _h(step):
    return None if step is None else step.handle


# Weight gradients of launches with at most this many voxels (batch included) run on the step context's side stream, beside
# the data-gradient chain (StepContext.side_stream); 0 = everything on one stream (the default).  Measured in round 5
# (profiles/r05s_ab_side_wgrad.txt, 160x192x160, one box, alternating): 0 -> 8.29 / 8.35 ms, 700 000 (levels 3-5 + CWM) ->
# 8.65 / 8.32, 1 300 000 -> 8.34, everything -> 8.34: no gain -- every fork is an extra cross-stream edge of the hipGraph (the
# host cost of a replay goes from 0.27 to 1.03 ms) and the 17 forks cost what the overlap of those 20-50 us kernels buys.
SIDE_WGRAD_MAX_VOXELS = float(os.environ.get("MODET_SIDE_WGRAD_MAX_VOXELS", "0"))


def conv3d_backward_weight(x, dy, want_bias, y_act=None, w=None, b=None, step=None, amax=None, norm=None, x_amax=None):
    """d_w, d_bias; with y_act (ConvBlock 1 -> 4 only) dy is the gradient w.r.t. LeakyReLU(conv) and the activation's
    derivative is applied while loading it.  With a StepContext (given, or bound to this thread) whose ``deferred()`` scope
    knows destinations for the parameters ``w`` / ``b``, the gradients go straight there at the scope's flush and
    (None, None) is returned.  amax: one-float tensor >= max |dy| (see _tag_amax) and the caller's word that x is an
    activation: the z-marching kernel then runs on two f16 pieces.  norm = (mean, rstd): x is a RAW ConvInsBlock output,
    normalised while the kernel stages it (modet_conv3d_bwd_weight_normin; only where modet_conv3d_bwd_weight_normin_ok).
    x_amax: the maxima of |x| on the device (see _tag_xamax) for an x that is NOT an activation: scaled by them instead."""
    _chk(x, dy)
    B, D, H, W, Cin = x.shape
    Cout = dy.shape[-1]
    L = _L()
    nb = L.modet_conv3d_bwd_weight_ws_bytes(B, D, H, W, Cin, Cout)
    n = float(B) * D * H * W
    scope = step if step is not None else current_step()
    dst = scope.destinations(w, b, want_bias) if scope is not None else None
    if dst is not None:
        dw, db = dst
        if SIDE_WGRAD_MAX_VOXELS and n <= SIDE_WGRAD_MAX_VOXELS and _TIMER is None:
            with torch.cuda.stream(scope.side_stream(x)):
                ws = _ws(nb, x)
                _lib.check(L.modet_conv3d_bwd_weight_defer(_p(x), _p(dy), _p(y_act), _p(dw), _p(db), _p(ws), nb, B, D, H, W, Cin,
                                                           Cout, _stream(), _h(scope)), "modet_conv3d_bwd_weight_defer")
            scope._keep.extend((x, dy, y_act))                # read by the side stream: alive until the flush joins it
        else:
            ws = _ws(nb, x)
            with _Guard(x, _conv_tag("wgrad", x.shape, Cin, Cout, f16=amax is not None and y_act is None), 54.0 * Cin * Cout * n,
                        4.0 * n * (Cin + Cout)):
                if norm is not None:
                    _lib.check(L.modet_conv3d_bwd_weight_normin(_p(x), _p(norm[0]), _p(norm[1]), _p(dy), _p(dw), _p(db), _p(ws), nb,
                                                                B, D, H, W, Cin, Cout, _p(amax), _stream(), _h(scope)),
                               "modet_conv3d_bwd_weight_normin")
                elif amax is not None and y_act is None:
                    _lib.check(L.modet_conv3d_bwd_weight_amax2(_p(x), _p(dy), _p(dw), _p(db), _p(ws), nb, B, D, H, W, Cin, Cout,
                                                               _p(amax), _p(x_amax), _stream(), _h(scope)), "modet_conv3d_bwd_weight_amax")
                else:
                    _lib.check(L.modet_conv3d_bwd_weight_defer(_p(x), _p(dy), _p(y_act), _p(dw), _p(db), _p(ws), nb, B, D, H, W, Cin,
                                                               Cout, _stream(), _h(scope)), "modet_conv3d_bwd_weight_defer")
        scope._keep.append(ws)                                # the partial tiles must survive until the flush
        if amax is not None:
            scope._keep.extend((amax, x_amax))                # (a queued launch reads the maxima at the flush, too)
        if y_act is None and L.modet_conv3d_wgrad_defers_operands(B, D, H, W, Cin, Cout):
            scope._keep.extend((x, dy))                       # small levels: the launch itself is queued and reads them at the flush
        scope.written.add(w.data_ptr())
        if db is not None:
            scope.written.add(b.data_ptr())
        return None, None
    ws = _ws(nb, x)
    dw = torch.empty((Cout, Cin, 3, 3, 3), dtype=torch.float32, device=x.device)
    db = torch.empty((Cout,), dtype=torch.float32, device=x.device) if want_bias else None
    with _Guard(x, _conv_tag("wgrad", x.shape, Cin, Cout, f16=amax is not None and y_act is None), 54.0 * Cin * Cout * n,
                4.0 * n * (Cin + Cout)):
        if norm is not None:
            _lib.check(L.modet_conv3d_bwd_weight_normin(_p(x), _p(norm[0]), _p(norm[1]), _p(dy), _p(dw), _p(db), _p(ws), nb, B, D, H,
                                                        W, Cin, Cout, _p(amax), _stream(), None), "modet_conv3d_bwd_weight_normin")
        elif y_act is not None:
            _lib.check(L.modet_conv3d_bwd_weight_act(_p(x), _p(dy), _p(y_act), _p(dw), _p(db), _p(ws), nb, B, D, H, W, Cin,
                                                     Cout, _stream()), "modet_conv3d_bwd_weight_act")
        elif amax is not None:
            _lib.check(L.modet_conv3d_bwd_weight_amax2(_p(x), _p(dy), _p(dw), _p(db), _p(ws), nb, B, D, H, W, Cin, Cout, _p(amax),
                                                       _p(x_amax), _stream(), None), "modet_conv3d_bwd_weight_amax")
        else:
            _lib.check(L.modet_conv3d_bwd_weight(_p(x), _p(dy), _p(dw), _p(db), _p(ws), nb, B, D, H, W, Cin, Cout,
                                                 _stream()), "modet_conv3d_bwd_weight")
    return dw, db


# ------------------------------------------------------------------------------------------------ autograd ops
class _Conv3d(Function):
    @staticmethod
    def forward(ctx, x, w, b, act, x_act=False):
        ctx.step = current_step()
        y = conv3d_forward(x, w, b, act, ctx.step, x_act)
        ctx.act = bool(act)
        ctx.has_bias = b is not None
        ctx.save_for_backward(x, w, y if act else None, b)
        return y

    @staticmethod
    def backward(ctx, dy):
        x, w, y, b = ctx.saved_tensors
        dy = dy.contiguous()
        if ctx.act and not ctx.needs_input_grad[0] and x.shape[-1] == 1 and w.shape[0] == 4:
            # first encoder block: no d_x, and the weight-gradient kernel folds LeakyReLU' into its d_y load
            dw, db = conv3d_backward_weight(x, dy, ctx.has_bias, y_act=y, w=w, b=b, step=ctx.step)
            return None, dw, db, None, None
        if ctx.act:
            g = torch.empty_like(dy)
            with _Guard(dy, "lrelu_bwd", dy.numel(), 12.0 * dy.numel()):
                _lib.check(_L().modet_lrelu_bwd(_p(dy), _p(y), _p(g), dy.numel(), _stream()), "modet_lrelu_bwd")
            dy = g
        # (the weight gradient first: on the small levels it goes to the side stream and runs beside the data gradient)
        amax = None if ctx.act else _amax_of(dy)                 # (x is whatever the caller convolved: no f16 weight gradient)
        dw, db = conv3d_backward_weight(x, dy, ctx.has_bias, w=w, b=b, step=ctx.step)
        dx = conv3d_backward_data(dy, w, x.shape[-1], ctx.step, amax) if ctx.needs_input_grad[0] else None
        return dx, dw, db, None, None


class _Conv3dStats(Function):
    """conv3d whose epilogue also produces InstanceNorm partial statistics (second output, not differentiable)"""

    @staticmethod
    def forward(ctx, x, w, b, x_act=False):
        _chk(x, w, b)
        ctx.step = current_step()
        ctx.x_act = bool(x_act)                 # x is an activation (bounded: see modet_conv3d_bwd_weight_amax)
        B, D, H, W, Cin = x.shape
        Cout = w.shape[0]
        L = _L()
        y = torch.empty((B, D, H, W, Cout), dtype=torch.float32, device=x.device)
        nb = L.modet_conv3d_ws_bytes(Cin, Cout)
        ws = _ws(nb, x)
        sb = L.modet_conv3d_stats_bytes(B, D, H, W, Cin, Cout)
        stats = torch.empty(sb // 4, dtype=torch.float32, device=x.device)
        n = float(B) * D * H * W
        # x is not an activation but its producer left max |x| on the device (the ConvBlock 1 -> 4 output, _tag_xamax): the f16
        # pieces are scaled by it (z-marching family; the others ignore it and run bf16x3)
        ctx.x_amax = None if ctx.x_act else _xamax_of(x)
        f16 = ctx.x_act or (ctx.x_amax is not None and L.modet_conv3d_kernel_family_v(B, D, H, W, Cin, Cout, 0, 3) == 2)
        with _Guard(x, _conv_tag("fwd", x.shape, Cin, Cout, 3, f16=f16), 54.0 * Cin * Cout * n, 4.0 * n * (Cin + Cout)):
            if ctx.x_amax is not None:
                _lib.check(L.modet_conv3d_fwd_stats_amax(_p(x), _p(w), _p(b), _p(y), _p(ws), nb, _p(stats), sb, B, D, H, W, Cin, Cout,
                                                         _p(ctx.x_amax), _stream(), _h(ctx.step)), "modet_conv3d_fwd_stats_amax")
            else:
                fn = L.modet_conv3d_fwd_stats_bounded if ctx.x_act else L.modet_conv3d_fwd_stats
                _lib.check(fn(_p(x), _p(w), _p(b), _p(y), _p(ws), nb, _p(stats), sb, B, D, H, W, Cin, Cout, _stream(), _h(ctx.step)),
                           "modet_conv3d_fwd_stats")
        ctx.has_bias = b is not None
        ctx.save_for_backward(x, w, b)
        ctx.mark_non_differentiable(stats)
        ctx.set_materialize_grads(False)        # no zeros_like(stats) fill launch for the statistics output's "gradient"
        return y, stats

    @staticmethod
    def backward(ctx, dy, _dstats):
        if dy is None:
            return None, None, None, None
        x, w, b = ctx.saved_tensors
        dy = dy.contiguous()
        amax = _amax_of(dy)
        dw, db = conv3d_backward_weight(x, dy, ctx.has_bias, w=w, b=b, step=ctx.step,
                                        amax=amax if (ctx.x_act or ctx.x_amax is not None) else None, x_amax=ctx.x_amax)
        dx = conv3d_backward_data(dy, w, x.shape[-1], ctx.step, amax) if ctx.needs_input_grad[0] else None
        return dx, dw, db, None


def _fuse_stats(x, w, needs_grad=None):
    """fuse the InstanceNorm statistics into this conv's epilogue?  The staged epilogue of the 16-wide configurations
    (Cout 4/8/16) carries them for free; in the direct-store configurations they cost the conv ~3 % and only pay off when
    no backward pass follows (measured: training +0.07 ms, inference -0.05 ms)."""
    B, D, H, W, Cin = x.shape
    Cout = w.shape[0]
    L = _L()
    if L.modet_conv3d_stats_bytes(B, D, H, W, Cin, Cout) == 0:
        return False
    if L.modet_conv3d_kernel_family(B, D, H, W, Cin, Cout, 0) in (1, 2, 5):  # the bf16x3 kernels carry the statistics at no cost
        return True
    if needs_grad is None:     # (inside an autograd.Function.forward grad mode is off: such callers pass their ctx.needs_input_grad)
        needs_grad = torch.is_grad_enabled() and (x.requires_grad or w.requires_grad)
    return Cout in (4, 8, 16) or not needs_grad


def conv3d_instnorm_lrelu(x, w, b, eps=1e-5, x_act=False):
    """ConvInsBlock = conv + InstanceNorm3d + LeakyReLU(0.1) (reference models.py:135-151); the norm statistics are
    fused into the conv epilogue when the configuration supports it.  x_act: see conv3d_forward"""
    if _fuse_stats(x, w):
        y, stats = _Conv3dStats.apply(x, w, b, x_act)
        return _InstNormLReLU.apply(y, eps, stats)
    return _InstNormLReLU.apply(_Conv3d.apply(x, w, b, False, x_act), eps, None)


def lazy_instnorm_conv3d(x_raw, stats_in, w, b, eps=1e-5, want_stats=True):
    """conv3d(LeakyReLU(InstanceNorm(x_raw)), w, b) -> (z_raw, z_stats or None); want_stats=False when z is not normalised.

    Without gradients (inference) the normalised tensor is never materialised: the statistics come from the previous
    conv's epilogue (or one statistics pass) and the conv kernel normalises its input tile while staging it (-3.5 % on the
    forward pass).  With gradients the tensor is needed by the weight gradient anyway (normalising on the fly there was
    measured slower), so the block runs as InstanceNorm + conv."""
    Cin = x_raw.shape[-1]
    needs_grad = torch.is_grad_enabled() and (x_raw.requires_grad or w.requires_grad)
    if not needs_grad and Cin % 4 == 0 and Cin > 1:
        mean, rstd = instnorm_stats(x_raw, stats_in, eps)
        return conv3d_forward_normin(x_raw, mean, rstd, w, b, want_stats)
    return _InstNormConv.apply(x_raw, stats_in, w, b, eps, want_stats)


# The data gradient of a conv whose input is LeakyReLU(InstanceNorm(x_raw)) can form that norm's backward statistics in
# its own epilogue (modet_conv3d_bwd_data_instats: kernel family 2 only): the norm's backward then skips its first pass
# over (d_y, x_raw).  False = the two ops run back to back as before (A/B switch of the parity tests).
FUSE_IN_DGRAD = os.environ.get("MODET_FUSE_IN_DGRAD", "1") != "0"
# Round 5: where the weight-gradient kernel can normalise its x operand while staging it (modet_conv3d_bwd_weight_normin_ok: the
# z-marching kernel, i.e. the level-1 layers) the TRAINING step does not materialise LeakyReLU(InstanceNorm(x_raw)) either: forward
# = statistics finalize + modet_conv3d_fwd_normin, weight gradient = modet_conv3d_bwd_weight_normin, data gradient and the norm's
# backward as before (they never needed the normalised tensor).  Saves the apply pass (0.12 ms at level 1) and 315 MB.
# False = materialise as before (A/B switch).
LAZY_IN_TRAIN = os.environ.get("MODET_LAZY_IN_TRAIN", "1") != "0"


class _InstNormConv(Function):
    """z = conv3d(LeakyReLU(InstanceNorm(x_raw)), w, b) as ONE autograd node (ConvInsBlock -> conv, reference
    models.py:186-219 / :252-254), so that its backward can hand the conv's data gradient and the norm's backward
    statistics over inside one kernel.  Returns (z, z_stats or None)."""

    @staticmethod
    def forward(ctx, x_raw, stats_in, w, b, eps, want_stats):
        _chk(x_raw, w, b)
        ctx.step = current_step()
        B, C = x_raw.shape[0], x_raw.shape[-1]
        V = x_raw.numel() // (B * C)
        L = _L()
        ctx.lazy = False
        if (LAZY_IN_TRAIN and x_raw.dim() == 5 and C % 4 == 0 and
                L.modet_conv3d_bwd_weight_normin_ok(B, x_raw.shape[1], x_raw.shape[2], x_raw.shape[3], C, w.shape[0])):
            mean, rstd = instnorm_stats(x_raw, stats_in, eps)
            z, stats = conv3d_forward_normin(x_raw, mean, rstd, w, b, want_stats, ctx.step)
            if stats is not None:
                ctx.mark_non_differentiable(stats)
            ctx.set_materialize_grads(False)
            ctx.has_bias = b is not None
            ctx.lazy = True
            ctx.save_for_backward(x_raw, mean, rstd, None, w, b)
            return z, stats
        y = torch.empty_like(x_raw)
        mean = torch.empty(B * C, dtype=torch.float32, device=x_raw.device)
        rstd = torch.empty_like(mean)
        with _Guard(x_raw, "instnorm_lrelu_fwd", 8.0 * x_raw.numel(), 8.0 * x_raw.numel()):
            if stats_in is not None:
                _lib.check(L.modet_instnorm_lrelu_fwd_stats(_p(x_raw), _p(y), _p(mean), _p(rstd), _p(stats_in),
                                                            stats_in.numel() * 4, B, V, C, eps, _stream()),
                           "modet_instnorm_lrelu_fwd_stats")
            else:
                nb = L.modet_instnorm_ws_bytes(B, V, C)
                ws = _ws(nb, x_raw)
                _lib.check(L.modet_instnorm_lrelu_fwd(_p(x_raw), _p(y), _p(mean), _p(rstd), _p(ws), nb, B, V, C, eps,
                                                      _stream()), "modet_instnorm_lrelu_fwd")
        _, D, H, W, Cin = y.shape
        Cout = w.shape[0]
        stats = None
        if want_stats and _fuse_stats(y, w, needs_grad=any(ctx.needs_input_grad)):
            z = torch.empty((B, D, H, W, Cout), dtype=torch.float32, device=y.device)
            nb = L.modet_conv3d_ws_bytes(Cin, Cout)
            ws = _ws(nb, y)
            sb = L.modet_conv3d_stats_bytes(B, D, H, W, Cin, Cout)
            stats = torch.empty(sb // 4, dtype=torch.float32, device=y.device)
            n = float(B) * D * H * W
            with _Guard(y, _conv_tag("fwd", y.shape, Cin, Cout, 3, f16=True), 54.0 * Cin * Cout * n, 4.0 * n * (Cin + Cout)):
                _lib.check(L.modet_conv3d_fwd_stats_bounded(_p(y), _p(w), _p(b), _p(z), _p(ws), nb, _p(stats), sb, B, D, H, W, Cin,
                                                            Cout, _stream(), _h(ctx.step)), "modet_conv3d_fwd_stats")    # (y: LeakyReLU(InstanceNorm(.)))
            ctx.mark_non_differentiable(stats)
        else:
            z = conv3d_forward(y, w, b, False, ctx.step, x_act=True)
        ctx.set_materialize_grads(False)        # no zeros_like(stats) fill launch for the statistics output's "gradient"
        ctx.has_bias = b is not None
        ctx.save_for_backward(x_raw, mean, rstd, y, w, b)
        return z, stats

    @staticmethod
    def backward(ctx, dz, _dstats):
        if dz is None:
            return None, None, None, None, None, None
        x_raw, mean, rstd, y, w, b = ctx.saved_tensors
        dz = dz.contiguous()
        B, D, H, W, C = x_raw.shape
        Cout = w.shape[0]
        V = D * H * W
        L = _L()
        d_raw = None
        amax = _amax_of(dz)
        if ctx.lazy:
            dw, db = conv3d_backward_weight(x_raw, dz, ctx.has_bias, w=w, b=b, step=ctx.step, amax=amax, norm=(mean, rstd))
        else:
            dw, db = conv3d_backward_weight(y, dz, ctx.has_bias, w=w, b=b, step=ctx.step, amax=amax)   # (first: see _Conv3d.backward)
        if ctx.needs_input_grad[0]:
            d_raw = torch.empty_like(x_raw)
            amax_out = _new_amax(x_raw)
            rb = L.modet_conv3d_bwd_data_instats_bytes(B, D, H, W, C, Cout) if FUSE_IN_DGRAD else 0
            if rb > 0:
                d_y = torch.empty_like(x_raw)
                rows = torch.empty(rb // 4, dtype=torch.float32, device=x_raw.device)
                nb = L.modet_conv3d_ws_bytes(C, Cout)
                ws = _ws(nb, dz)
                n = float(B) * V
                with _Guard(dz, _conv_tag("dgrad", dz.shape, C, Cout, f16=amax is not None), 54.0 * C * Cout * n, 4.0 * n * (2 * C + Cout)):
                    _lib.check(L.modet_conv3d_bwd_data_instats_amax(_p(dz), _p(w), _p(d_y), _p(x_raw), _p(mean), _p(rstd), _p(rows),
                                                                    rb, _p(ws), nb, B, D, H, W, C, Cout, _p(amax), _stream(),
                                                                    _h(ctx.step)), "modet_conv3d_bwd_data_instats")
                nb2 = 2 * B * C * 4
                ws2 = _ws(nb2, x_raw)
                with _Guard(x_raw, "instnorm_lrelu_bwd", 7.0 * x_raw.numel(), 12.0 * x_raw.numel()):
                    _lib.check(L.modet_instnorm_lrelu_bwd_rows_amax(_p(d_y), _p(x_raw), _p(mean), _p(rstd), _p(d_raw), _p(rows), rb,
                                                                    _p(ws2), nb2, B, V, C, _p(amax_out), _stream()),
                               "modet_instnorm_lrelu_bwd_rows")
            else:
                d_y = conv3d_backward_data(dz, w, C, ctx.step, amax)
                nb = L.modet_instnorm_ws_bytes(B, V, C)
                ws = _ws(nb, x_raw)
                with _Guard(x_raw, "instnorm_lrelu_bwd", 14.0 * x_raw.numel(), 12.0 * x_raw.numel()):
                    _lib.check(L.modet_instnorm_lrelu_bwd_amax(_p(d_y), _p(x_raw), _p(mean), _p(rstd), _p(d_raw), _p(ws), nb, B, V, C,
                                                               _p(amax_out), _stream()), "modet_instnorm_lrelu_bwd")
            _tag_amax(d_raw, amax_out)
        return d_raw, None, dw, db, None, None


def conv3d_with_stats(x, w, b, x_act=False):
    """raw 3x3x3 conv output plus, when the configuration supports it, the InstanceNorm partial statistics from its
    epilogue (else None).  x_act: the caller's word that x is an activation (a ConvBlock / ConvInsBlock output or a pooled
    one), which lets the weight gradient split it into f16 pieces with a fixed scale."""
    if _fuse_stats(x, w):
        return _Conv3dStats.apply(x, w, b, x_act)
    return _Conv3d.apply(x, w, b, False, x_act), None


def conv3d(x, w, b=None, act=False, x_act=False):
    """3x3x3 conv, zero pad 1 (+ fused LeakyReLU(0.1) if act).  reference: nn.Conv3d, models.py:127,:144,:254.  Any input range
    (x_act=True: the caller's word that |x| < 4 094 -- half the matrix work, see conv3d_forward)"""
    return _Conv3d.apply(x, w, b, act, x_act)


class _InstNormLReLU(Function):
    @staticmethod
    def forward(ctx, x, eps, stats=None):
        _chk(x)
        B, C = x.shape[0], x.shape[-1]
        V = x.numel() // (B * C)
        y = torch.empty_like(x)
        mean = torch.empty(B * C, dtype=torch.float32, device=x.device)
        rstd = torch.empty_like(mean)
        L = _L()
        if stats is not None:
            with _Guard(x, "instnorm_lrelu_fwd", 8.0 * x.numel(), 8.0 * x.numel()):
                _lib.check(L.modet_instnorm_lrelu_fwd_stats(_p(x), _p(y), _p(mean), _p(rstd), _p(stats), stats.numel() * 4,
                                                            B, V, C, eps, _stream()), "modet_instnorm_lrelu_fwd_stats")
        else:
            nb = L.modet_instnorm_ws_bytes(B, V, C)
            ws = _ws(nb, x)
            with _Guard(x, "instnorm_lrelu_fwd", 8.0 * x.numel(), 8.0 * x.numel()):
                _lib.check(L.modet_instnorm_lrelu_fwd(_p(x), _p(y), _p(mean), _p(rstd), _p(ws), nb, B, V, C, eps,
                                                      _stream()), "modet_instnorm_lrelu_fwd")
        ctx.save_for_backward(x, mean, rstd)
        return y

    @staticmethod
    def backward(ctx, dy):
        x, mean, rstd = ctx.saved_tensors
        dy = dy.contiguous()
        B, C = x.shape[0], x.shape[-1]
        V = x.numel() // (B * C)
        dx = torch.empty_like(x)
        amax = _new_amax(x)
        L = _L()
        nb = L.modet_instnorm_ws_bytes(B, V, C)
        ws = _ws(nb, x)
        with _Guard(x, "instnorm_lrelu_bwd", 14.0 * x.numel(), 12.0 * x.numel()):
            _lib.check(L.modet_instnorm_lrelu_bwd_amax(_p(dy), _p(x), _p(mean), _p(rstd), _p(dx), _p(ws), nb, B, V, C, _p(amax),
                                                       _stream()), "modet_instnorm_lrelu_bwd")
        return _tag_amax(dx, amax), None, None


def instnorm_lrelu(x, eps=1e-5):
    """InstanceNorm3d(affine=False) + LeakyReLU(0.1), channels-last.  reference: models.py:144-150"""
    return _InstNormLReLU.apply(x, eps, None)


class _AvgPool2(Function):
    @staticmethod
    def forward(ctx, x):
        _chk(x)
        B, D, H, W, C = x.shape
        y = torch.empty((B, D // 2, H // 2, W // 2, C), dtype=torch.float32, device=x.device)
        with _Guard(x, "avgpool2_fwd", x.numel(), 4.5 * x.numel()):
            _lib.check(_L().modet_avgpool2_fwd(_p(x), _p(y), B, D, H, W, C, _stream()), "modet_avgpool2_fwd")
        ctx.shape = (B, D, H, W, C)
        return y

    @staticmethod
    def backward(ctx, dy):
        dy = dy.contiguous()
        B, D, H, W, C = ctx.shape
        dx = torch.empty(ctx.shape, dtype=torch.float32, device=dy.device)
        with _Guard(dy, "avgpool2_bwd", dx.numel(), 4.5 * dx.numel()):
            _lib.check(_L().modet_avgpool2_bwd(_p(dy), None, _p(dx), B, D, H, W, C, _stream()), "modet_avgpool2_bwd")
        return dx


def avgpool2(x):
    """nn.AvgPool3d(2), channels-last.  reference: models.py:201,:207,:213,:219"""
    return _AvgPool2.apply(x)


class _PoolTee(Function):
    """x -> (avgpool2(x), x): the pyramid feeds every feature map both to the next level's pooling and to the
    attention/warp branch; one backward kernel returns g_x + unpool(g_pooled)/8 instead of a pool-backward pass
    followed by autograd's add over two full-size tensors."""

    @staticmethod
    def forward(ctx, x):
        _chk(x)
        B, D, H, W, C = x.shape
        y = torch.empty((B, D // 2, H // 2, W // 2, C), dtype=torch.float32, device=x.device)
        with _Guard(x, "avgpool2_fwd", x.numel(), 4.5 * x.numel()):
            _lib.check(_L().modet_avgpool2_fwd(_p(x), _p(y), B, D, H, W, C, _stream()), "modet_avgpool2_fwd")
        ctx.shape = (B, D, H, W, C)
        return y, x.view_as(x)

    @staticmethod
    def backward(ctx, gy, gx):
        B, D, H, W, C = ctx.shape
        if gy is None:
            return gx
        gy = gy.contiguous()
        add = None if gx is None else gx.contiguous()
        dx = torch.empty(ctx.shape, dtype=torch.float32, device=gy.device)
        with _Guard(gy, "avgpool2_bwd", dx.numel(), 4.5 * dx.numel() + (4.0 * dx.numel() if add is not None else 0.0)):
            _lib.check(_L().modet_avgpool2_bwd(_p(gy), _p(add), _p(dx), B, D, H, W, C, _stream()), "modet_avgpool2_bwd")
        return dx


def pool_tee(x):
    """(avgpool2(x), x) with a fused backward; see _PoolTee"""
    return _PoolTee.apply(x)


class _PoolTeeSplit(Function):
    """x (2B,...) -> (avgpool2(x), x[:B], x[B:]): _PoolTee for the [moving; fixed] batch whose two halves go to
    different consumers (warp / projection).  The backward runs the fused unpool+add once per half, each with its own
    addend, so the two half gradients are never concatenated."""

    @staticmethod
    def forward(ctx, x, Bh):
        _chk(x)
        B, D, H, W, C = x.shape
        y = torch.empty((B, D // 2, H // 2, W // 2, C), dtype=torch.float32, device=x.device)
        with _Guard(x, "avgpool2_fwd", x.numel(), 4.5 * x.numel()):
            _lib.check(_L().modet_avgpool2_fwd(_p(x), _p(y), B, D, H, W, C, _stream()), "modet_avgpool2_fwd")
        ctx.shape = (B, D, H, W, C)
        ctx.Bh = Bh
        return y, x[:Bh], x[Bh:]

    @staticmethod
    def backward(ctx, gy, ga, gb):
        B, D, H, W, C = ctx.shape
        Bh = ctx.Bh
        dx = torch.empty(ctx.shape, dtype=torch.float32, device=(gy if gy is not None else ga if ga is not None else gb).device)
        if gy is None:
            for sl, g in ((slice(0, Bh), ga), (slice(Bh, B), gb)):
                if g is None:
                    dx[sl].zero_()
                else:
                    dx[sl].copy_(g)
            return dx, None
        gy = gy.contiguous()
        L = _L()
        with _Guard(gy, "avgpool2_bwd", dx.numel(), 8.5 * dx.numel()):
            for lo, hi, g in ((0, Bh, ga), (Bh, B, gb)):
                add = None if g is None else g.contiguous()
                _lib.check(L.modet_avgpool2_bwd(_p(gy[lo:hi]), _p(add), _p(dx[lo:hi]), hi - lo, D, H, W, C, _stream()),
                           "modet_avgpool2_bwd")
        return dx, None


def pool_tee_split(x, Bh):
    """(avgpool2(x), x[:Bh], x[Bh:]) with a fused backward; see _PoolTeeSplit"""
    return _PoolTeeSplit.apply(x, Bh)


class _InstNormLReLUPoolSplit(Function):
    """_InstNormLReLU followed by _PoolTeeSplit as ONE node: x_raw (2B,...) -> (avgpool2(y), y[:B], y[B:]) with
    y = LeakyReLU(InstanceNorm(x_raw)).  Forward: statistics (finalize only, or one pass), then one kernel that writes y
    and its pooled copy (modet_instnorm_lrelu_apply_pool: the separate AvgPool3d re-read y).  Backward: the two nodes'
    backward kernels in sequence, unchanged.  Bit-identical to the two-node form."""

    @staticmethod
    def forward(ctx, x, eps, stats, Bh):
        _chk(x)
        B, D, H, W, C = x.shape
        V = D * H * W
        L = _L()
        mean = torch.empty(B * C, dtype=torch.float32, device=x.device)
        rstd = torch.empty_like(mean)
        y = torch.empty_like(x)
        pooled = torch.empty((B, D // 2, H // 2, W // 2, C), dtype=torch.float32, device=x.device)
        with _Guard(x, "instnorm_lrelu_fwd", 9.0 * x.numel(), 8.5 * x.numel()):
            if stats is not None:
                _lib.check(L.modet_instnorm_stats(_p(x), _p(mean), _p(rstd), _p(stats), stats.numel() * 4, None, 0, B, V, C,
                                                  eps, _stream()), "modet_instnorm_stats")
            else:
                nb = L.modet_instnorm_ws_bytes(B, V, C)
                ws = _ws(nb, x)
                _lib.check(L.modet_instnorm_stats(_p(x), _p(mean), _p(rstd), None, 0, _p(ws), nb, B, V, C, eps, _stream()),
                           "modet_instnorm_stats")
            _lib.check(L.modet_instnorm_lrelu_apply_pool(_p(x), _p(mean), _p(rstd), _p(y), _p(pooled), B, D, H, W, C,
                                                         _stream()), "modet_instnorm_lrelu_apply_pool")
        ctx.save_for_backward(x, mean, rstd)
        ctx.Bh = Bh
        ctx.set_materialize_grads(False)        # an unused output arrives as None (no zeros fill launch); handled below
        return pooled, y[:Bh], y[Bh:]

    @staticmethod
    def backward(ctx, gy, ga, gb):
        if not ctx.needs_input_grad[0] or (gy is None and ga is None and gb is None):
            return None, None, None, None
        x, mean, rstd = ctx.saved_tensors
        B, D, H, W, C = x.shape
        Bh = ctx.Bh
        V = D * H * W
        L = _L()
        dx = torch.empty_like(x)
        amax = _new_amax(x)
        nb = L.modet_instnorm_ws_bytes(B, V, C)
        ws = _ws(nb, x)
        if gy is not None:
            # d_y = unpool(gy) / 8 + [ga ; gb] is formed inside the two InstanceNorm backward passes, never written
            gy = gy.contiguous()
            ga = None if ga is None else ga.contiguous()
            gb = None if gb is None else gb.contiguous()
            with _Guard(x, "instnorm_lrelu_bwd", 15.0 * x.numel(), 12.5 * x.numel()):
                _lib.check(L.modet_instnorm_lrelu_bwd_pool_amax(_p(gy), _p(ga), _p(gb), Bh, _p(x), _p(mean), _p(rstd), _p(dx), _p(ws),
                                                                nb, B, D, H, W, C, _p(amax), _stream()),
                           "modet_instnorm_lrelu_bwd_pool")
            return _tag_amax(dx, amax), None, None, None
        dy = torch.empty_like(x)
        for sl, g in ((slice(0, Bh), ga), (slice(Bh, B), gb)):
            if g is None:
                dy[sl].zero_()
            else:
                dy[sl].copy_(g)
        with _Guard(x, "instnorm_lrelu_bwd", 14.0 * x.numel(), 12.0 * x.numel()):
            _lib.check(L.modet_instnorm_lrelu_bwd_amax(_p(dy), _p(x), _p(mean), _p(rstd), _p(dx), _p(ws), nb, B, V, C, _p(amax),
                                                       _stream()), "modet_instnorm_lrelu_bwd")
        return _tag_amax(dx, amax), None, None, None


def instnorm_lrelu_pool_tee_split(x_raw, stats, Bh, eps=1e-5):
    """(avgpool2(y), y[:Bh], y[Bh:]) for y = LeakyReLU(InstanceNorm(x_raw)); see _InstNormLReLUPoolSplit"""
    return _InstNormLReLUPoolSplit.apply(x_raw, eps, stats, Bh)


class _ProjLN(Function):
    @staticmethod
    def forward(ctx, x, Wt, b, gamma, beta, eps):
        _chk(x, Wt, b, gamma, beta)
        Cin = x.shape[-1]
        dim = Wt.shape[0]
        N = x.numel() // Cin
        y = torch.empty(x.shape[:-1] + (dim,), dtype=torch.float32, device=x.device)
        with _Guard(x, f"proj_ln_fwd[{Cin}->{dim}]", N * (2.0 * Cin * dim + 8.0 * dim), 4.0 * N * (Cin + dim)):
            _lib.check(_L().modet_proj_ln_fwd(_p(x), _p(Wt), _p(b), _p(gamma), _p(beta), _p(y), N, Cin, dim, eps,
                                              _stream()), "modet_proj_ln_fwd")
        ctx.save_for_backward(x, Wt, b, gamma, beta)
        ctx.eps = eps
        ctx.step = current_step()
        return y

    @staticmethod
    def backward(ctx, dy):
        x, Wt, b, gamma, beta = ctx.saved_tensors
        dy = dy.contiguous()
        Cin = x.shape[-1]
        dim = Wt.shape[0]
        N = x.numel() // Cin
        dx = torch.empty_like(x)
        dst = ctx.step.claim(Wt, b, gamma, beta) if ctx.step is not None else None
        dW, db, dg, dbeta = dst if dst is not None else (torch.empty_like(Wt), torch.empty_like(b), torch.empty_like(gamma),
                                                         torch.empty_like(gamma))
        L = _L()
        nb = L.modet_proj_ln_bwd_ws_bytes(N, Cin, dim)
        ws = _ws(nb, x)
        with _Guard(x, f"proj_ln_bwd[{Cin}->{dim}]", N * (6.0 * Cin * dim + 20.0 * dim), 4.0 * N * (2 * Cin + dim)):
            _lib.check(L.modet_proj_ln_bwd(_p(x), _p(Wt), _p(b), _p(gamma), _p(dy), _p(dx), _p(dW), _p(db), _p(dg),
                                           _p(dbeta), _p(ws), nb, N, Cin, dim, ctx.eps, _stream()), "modet_proj_ln_bwd")
        if dst is not None:
            return dx, None, None, None, None, None
        return dx, dW, db, dg, dbeta, None


class _ProjLNPair(Function):
    """The projection layer applied to two inputs of one shape with the same parameters; the backward sums the
    parameter gradients of both uses in one reduction (modet_proj_ln_bwd_pair) instead of autograd adding two results."""

    @staticmethod
    def forward(ctx, x1, x2, Wt, b, gamma, beta, eps):
        _chk(x1, x2, Wt, b, gamma, beta)
        Cin = x1.shape[-1]
        dim = Wt.shape[0]
        N = x1.numel() // Cin
        L = _L()
        ys = [torch.empty(x.shape[:-1] + (dim,), dtype=torch.float32, device=x.device) for x in (x1, x2)]
        with _Guard(x1, f"proj_ln_fwd[{Cin}->{dim}]", 2 * N * (2.0 * Cin * dim + 8.0 * dim), 8.0 * N * (Cin + dim)):
            _lib.check(L.modet_proj_ln_fwd_pair(_p(x1), _p(x2), _p(Wt), _p(b), _p(gamma), _p(beta), _p(ys[0]), _p(ys[1]), N, Cin, dim,
                                                eps, _stream()), "modet_proj_ln_fwd_pair")
        ctx.save_for_backward(x1, x2, Wt, b, gamma, beta)
        ctx.eps = eps
        ctx.step = current_step()
        return ys[0], ys[1]

    @staticmethod
    def backward(ctx, dy1, dy2):
        x1, x2, Wt, b, gamma, beta = ctx.saved_tensors
        dy1, dy2 = dy1.contiguous(), dy2.contiguous()
        Cin = x1.shape[-1]
        dim = Wt.shape[0]
        N = x1.numel() // Cin
        dx1, dx2 = torch.empty_like(x1), torch.empty_like(x2)
        dst = ctx.step.claim(Wt, b, gamma, beta) if ctx.step is not None else None
        dW, db, dg, dbeta = dst if dst is not None else (torch.empty_like(Wt), torch.empty_like(b), torch.empty_like(gamma),
                                                         torch.empty_like(gamma))
        L = _L()
        nb = L.modet_proj_ln_bwd_pair_ws_bytes(N, Cin, dim)
        ws = _ws(nb, x1)
        later = dst is not None and DEFER_LEAF_REDUCTIONS
        out4 = (None, None, None, None) if later else (dW, db, dg, dbeta)
        with _Guard(x1, f"proj_ln_bwd[{Cin}->{dim}]", 2 * N * (6.0 * Cin * dim + 20.0 * dim), 8.0 * N * (2 * Cin + dim)):
            _lib.check(L.modet_proj_ln_bwd_pair(_p(x1), _p(dy1), _p(dx1), _p(x2), _p(dy2), _p(dx2), _p(Wt), _p(b), _p(gamma),
                                                _p(out4[0]), _p(out4[1]), _p(out4[2]), _p(out4[3]), _p(ws), nb, N, Cin, dim, ctx.eps,
                                                _stream()), "modet_proj_ln_bwd_pair")
        if later:
            _defer_proj(ctx.step, ws, N, Cin, dim, dW, db, dg, dbeta)
        if dst is not None:
            return dx1, dx2, None, None, None, None, None
        return dx1, dx2, dW, db, dg, dbeta, None


def proj_ln_pair(x1, x2, Wt, b, gamma, beta, eps=1e-5):
    """(proj_ln(x1), proj_ln(x2)) with shared parameters -- ModeT's q / k projections of a level (models.py:371-372)"""
    if (x1.shape == x2.shape and torch.is_grad_enabled() and x1.requires_grad and x2.requires_grad and
            _L().modet_proj_ln_bwd_pair_ws_bytes(x1.numel() // x1.shape[-1], x1.shape[-1], Wt.shape[0]) > 0):
        return _ProjLNPair.apply(x1, x2, Wt, b, gamma, beta, eps)
    return _ProjLN.apply(x1, Wt, b, gamma, beta, eps), _ProjLN.apply(x2, Wt, b, gamma, beta, eps)


def proj_ln(x, Wt, b, gamma, beta, eps=1e-5):
    """Linear + LayerNorm on channels-last voxels.  reference: ProjectionLayer, models.py:230-241"""
    return _ProjLN.apply(x, Wt, b, gamma, beta, eps)


class _NA(Function):
    @staticmethod
    def forward(ctx, q, k, rpb, heads, scale):
        _chk(q, k, rpb)
        B, D, H, W, C = q.shape
        if k.shape != q.shape:
            raise RuntimeError("neighbourhood attention: q and k shapes differ")
        hd = C // heads
        out = torch.empty((B, D, H, W, heads * 3), dtype=torch.float32, device=q.device)
        need_grad = any(ctx.needs_input_grad[:3])
        lse = torch.empty((B, D, H, W, heads), dtype=torch.float32, device=q.device) if need_grad else None
        nvh = float(B) * D * H * W * heads      # 60 B and ~620 flop per voxel-head
        with _Guard(q, f"na_fwd[h{heads}]", 620.0 * nvh, 60.0 * nvh):
            _lib.check(_L().modet_na_fwd(_p(q), _p(k), _p(rpb), _p(out), _p(lse), B, D, H, W, heads, hd, float(scale),
                                         _stream()), "modet_na_fwd")
        if need_grad:
            ctx.save_for_backward(q, k, rpb, out, lse)
        ctx.heads, ctx.scale = heads, float(scale)
        ctx.step = current_step()
        return out

    @staticmethod
    def backward(ctx, dout):
        q, k, rpb, out, lse = ctx.saved_tensors
        dout = dout.contiguous()
        B, D, H, W, C = q.shape
        heads = ctx.heads
        dq, dk = torch.empty_like(q), torch.empty_like(k)
        dst = ctx.step.claim(rpb) if ctx.step is not None else None
        drpb = dst[0] if dst is not None else torch.empty_like(rpb)
        L = _L()
        nb = L.modet_na_bwd_ws_bytes(B, D, H, W, heads)
        ws = _ws(nb, q)
        nvh = float(B) * D * H * W * heads      # reads q,k,d_out,out,lse (19 floats), writes d_q,d_k (12)
        later = dst is not None and DEFER_LEAF_REDUCTIONS
        with _Guard(q, f"na_bwd[h{heads}]", 1900.0 * nvh, 124.0 * nvh):
            _lib.check(L.modet_na_bwd(_p(q), _p(k), _p(rpb), _p(out), _p(lse), _p(dout), _p(dq), _p(dk), _p(None if later else drpb),
                                      _p(ws), nb, B, D, H, W, heads, C // heads, ctx.scale, _stream()), "modet_na_bwd")
        if later:
            _defer_rpb(ctx.step, ws, B, D, H, W, heads, C // heads, drpb)
        return dq, dk, (None if dst is not None else drpb), None, None


class _LevelAttnBF16(Function):
    """One pyramid level's matching step with bf16 STORAGE of its internal tensors (BASELINE.json configs[4]):
        Mw = warp(M, flow) (or M itself at the coarsest level)  ->  q = proj_ln(F), k = proj_ln(Mw)  ->  out = NA(q, k, rpb)
    as ONE autograd node, so that Mw, q and k -- which no other op reads -- live in HBM as bf16 and never cross an autograd edge
    (autograd would cast fp32 gradients arriving at a bf16 tensor; here every gradient stays fp32).  All arithmetic is the fp32
    kernels': a bf16 input is widened on load, a bf16 output rounded to nearest even on store
    (modet_warp_fwd_o16, modet_proj_ln_fwd_t, modet_na_fwd_t / _bwd_t, modet_proj_ln_bwd_pair_t).
    reference: ModeT/models.py:371-376 (projection + attention of a level), :55-67 (the warp in front of it)."""

    @staticmethod
    def forward(ctx, F, M, flow, Wt, b, gamma, beta, rpb, heads, scale, eps, tee=False):
        _chk(Wt, b, gamma, beta, rpb)
        ctx.tee = bool(tee) and flow is not None
        ctx.set_materialize_grads(False)
        ctx.step = current_step()
        ctx.beta_ref = beta                  # (only its address and shape: the key of its gradient destination)
        B, D, H, W, Cin = F.shape
        dim = Wt.shape[0]
        N = B * D * H * W
        L = _L()
        n = float(N)
        # level features handed over as fp32 handles with bf16 data (_InstNormLReLUBF16PoolSplit, features16): read the data
        # (the handles themselves own one element of storage expanded to the shape)
        Fd, Md = getattr(F, "data16", None), getattr(M, "data16", None)
        Fd = F if Fd is None else Fd
        Md = M if Md is None else Md
        _chk16(Fd, Md)
        m16 = int(Md.dtype == torch.bfloat16)
        if flow is not None:
            _chk(flow)
            Mw = torch.empty(M.shape, dtype=torch.bfloat16, device=M.device)
            with _Guard(M, f"warp_fwd[C{Cin}]", n * (24.0 * Cin + 30.0), n * ((2.0 if m16 else 4.0) * Cin + 2.0 * Cin + 12.0)):
                _lib.check(L.modet_warp_fwd_t(_p(Md), m16, _p(flow), _p(Mw), 1, B, D, H, W, Cin, _stream()), "modet_warp_fwd_t")
        else:
            Mw = Md
        q = torch.empty((B, D, H, W, dim), dtype=torch.bfloat16, device=F.device)
        k = torch.empty_like(q)
        for x, y in ((Fd, q), (Mw, k)):
            x16 = int(x.dtype == torch.bfloat16)
            with _Guard(F, f"proj_ln_fwd[{Cin}->{dim}]", n * (2.0 * Cin * dim + 8.0 * dim), n * ((2.0 if x16 else 4.0) * Cin + 2.0 * dim)):
                _lib.check(L.modet_proj_ln_fwd_t(_p(x), x16, _p(Wt), _p(b), _p(gamma), _p(beta), _p(y), 1, N, Cin, dim, eps, _stream()),
                           "modet_proj_ln_fwd_t")
        out = torch.empty((B, D, H, W, heads * 3), dtype=torch.float32, device=F.device)
        need_grad = any(ctx.needs_input_grad)
        lse = torch.empty((B, D, H, W, heads), dtype=torch.float32, device=F.device) if need_grad else None
        nvh = n * heads
        with _Guard(F, f"na_fwd[h{heads}]", 620.0 * nvh, 36.0 * nvh):
            _lib.check(L.modet_na_fwd_t(_p(q), _p(k), 1, _p(rpb), _p(out), _p(lse), B, D, H, W, heads, dim // heads, float(scale),
                                        _stream()), "modet_na_fwd_t")
        if need_grad:
            ctx.save_for_backward(Fd, Md, flow, Mw if flow is not None else None, q, k, Wt, b, gamma, rpb, out, lse)
        ctx.heads, ctx.scale, ctx.eps = heads, float(scale), eps
        if ctx.tee:                                        # the flow's second consumer takes this alias (see _WarpTee)
            return out, flow.view_as(flow)
        return out

    @staticmethod
    def backward(ctx, dout, galias=None):
        F, M, flow, Mw, q, k, Wt, b, gamma, rpb, out, lse = ctx.saved_tensors      # (F, M: the data, fp32 or bf16)
        if dout is None:                                   # (only the alias was used)
            return None, None, galias, None, None, None, None, None, None, None, None, None
        dout = dout.contiguous()
        galias = None if galias is None else galias.contiguous()
        B, D, H, W, Cin = F.shape
        dim = Wt.shape[0]
        heads = ctx.heads
        N = B * D * H * W
        n = float(N)
        L = _L()
        dq = torch.empty((B, D, H, W, dim), dtype=torch.float32, device=F.device)
        dk = torch.empty_like(dq)
        # the five parameter gradients straight into the flat gradient buffer when a deferred() scope offers destinations
        dst = ctx.step.claim(Wt, b, gamma, ctx.beta_ref, rpb) if ctx.step is not None else None
        drpb = dst[4] if dst is not None else torch.empty_like(rpb)
        nb = L.modet_na_bwd_ws_bytes(B, D, H, W, heads)
        ws = _ws(nb, F)
        nvh = n * heads
        later = dst is not None and DEFER_LEAF_REDUCTIONS
        with _Guard(F, f"na_bwd[h{heads}]", 1900.0 * nvh, 100.0 * nvh):
            _lib.check(L.modet_na_bwd_t(_p(q), _p(k), 1, _p(rpb), _p(out), _p(lse), _p(dout), _p(dq), _p(dk), _p(None if later else drpb),
                                        _p(ws), nb, B, D, H, W, heads, dim // heads, ctx.scale, _stream()), "modet_na_bwd_t")
        if later:
            _defer_rpb(ctx.step, ws, B, D, H, W, heads, dim // heads, drpb)
        x2 = Mw if flow is not None else M
        dF = torch.empty(F.shape, dtype=torch.float32, device=F.device)
        dMw = torch.empty(M.shape, dtype=torch.float32, device=M.device)
        dW, db, dg, dbeta = dst[:4] if dst is not None else (torch.empty_like(Wt), torch.empty_like(b), torch.empty_like(gamma),
                                                             torch.empty_like(gamma))
        nb2 = L.modet_proj_ln_bwd_pair_ws_bytes(N, Cin, dim)
        if nb2 == 0:
            raise RuntimeError(f"level attention (bf16): no paired projection backward for Cin {Cin}, dim {dim}")
        ws2 = _ws(nb2, F)
        with _Guard(F, f"proj_ln_bwd[{Cin}->{dim}]", 2 * n * (6.0 * Cin * dim + 20.0 * dim), n * (14.0 * Cin + 8.0 * dim)):
            out4 = (None, None, None, None) if later else (dW, db, dg, dbeta)
            _lib.check(L.modet_proj_ln_bwd_pair_t(_p(F), int(F.dtype == torch.bfloat16), _p(dq), _p(dF), _p(x2),
                                                  int(x2.dtype == torch.bfloat16), _p(dk), _p(dMw),
                                                  _p(Wt), _p(b), _p(gamma), _p(out4[0]), _p(out4[1]), _p(out4[2]), _p(out4[3]), _p(ws2),
                                                  nb2, N, Cin, dim, ctx.eps, _stream()), "modet_proj_ln_bwd_pair_t")
        if later:
            _defer_proj(ctx.step, ws2, N, Cin, dim, dW, db, dg, dbeta)
        if dst is not None:
            dW = db = dg = dbeta = drpb = None           # (written in place: nothing for autograd to hand on)
        if flow is None:
            return dF, dMw, None, dW, db, dg, dbeta, drpb, None, None, None, None
        dM = torch.empty(M.shape, dtype=torch.float32, device=M.device) if ctx.needs_input_grad[1] else None
        dflow = torch.empty_like(flow) if ctx.needs_input_grad[2] else None
        if dM is not None or dflow is not None:
            with _Guard(M, _warp_bwd_tag(M, dM, 0, 0), n * (60.0 * Cin + 40.0), 4.0 * n * (3 * Cin + 6)):
                _warp_backward(M, flow, dMw, dM, dflow, galias, 0, 0)
        elif galias is not None:
            dflow = galias
        return dF, dM, dflow, dW, db, dg, dbeta, drpb, None, None, None, None


def level_attention_bf16(F, M, flow, Wt, b, gamma, beta, rpb, heads, scale, eps=1e-5, tee=False):
    """NA(proj_ln(F), proj_ln(warp(M, flow))) with the warped features, q and k stored as bf16 (flow None: no warp); see
    _LevelAttnBF16.  Returns the expected offset (B, D, H, W, heads * 3), fp32; with ``tee`` (and a flow) the pair
    (offset, flow): the flow's other consumer takes the returned flow and its gradient is added inside the node's warp
    backward kernel (as ops.warp_tee)."""
    return _LevelAttnBF16.apply(F, M, flow, Wt, b, gamma, beta, rpb, heads, scale, eps, tee)


class _Corr3d(Function):
    @staticmethod
    def forward(ctx, mov, fix):
        _chk(mov, fix)
        B, D, H, W, C = mov.shape
        if fix.shape != mov.shape:
            raise RuntimeError("correlation3d: mov and fix shapes differ")
        L = _L()
        corr = torch.empty((B, 27, D, H, W), dtype=torch.float32, device=mov.device)
        nb = L.modet_corr3d_ws_bytes(B, D, H, W, C)
        ws = _ws(nb, mov)
        n = float(B) * D * H * W
        with _Guard(mov, f"corr3d_fwd[C{C}]", n * C * (54.0 + 54.0), 4.0 * n * (2 * C + 27)):
            _lib.check(L.modet_corr3d_fwd(_p(mov), _p(fix), _p(corr), _p(ws), nb, B, D, H, W, C, _stream()), "modet_corr3d_fwd")
        ctx.save_for_backward(mov, fix)
        return corr

    @staticmethod
    def backward(ctx, dcorr):
        mov, fix = ctx.saved_tensors
        dcorr = dcorr.contiguous()
        B, D, H, W, C = mov.shape
        L = _L()
        dmov, dfix = torch.empty_like(mov), torch.empty_like(fix)
        nb = L.modet_corr3d_ws_bytes(B, D, H, W, C)
        ws = _ws(nb, mov)
        n = float(B) * D * H * W
        with _Guard(mov, f"corr3d_bwd[C{C}]", n * C * 4 * 54.0, 4.0 * n * (4 * C + 27)):
            _lib.check(L.modet_corr3d_bwd(_p(mov), _p(fix), _p(dcorr), _p(dmov), _p(dfix), _p(ws), nb, B, D, H, W, C,
                                          _stream()), "modet_corr3d_bwd")
        return dmov, dfix


def correlation3d(mov, fix):
    """PR++ Correlation3D on channels-last features (B,D,H,W,C) -> (B,27,D,H,W); see csrc/corr3d.hip"""
    return _Corr3d.apply(mov, fix)


def neighbourhood_attention(q, k, rpb, heads, scale):
    """Fused ModeTransformer.forward: (B,D,H,W,heads*6) x2 -> (B,D,H,W,heads*3).  reference: models.py:308-334"""
    return _NA.apply(q, k, rpb.contiguous(), heads, scale)


class _Warp(Function):
    @staticmethod
    def forward(ctx, src, flow, mode, add_flow, flow_bound=0):
        _chk(src, flow)
        B, D, H, W, C = src.shape
        if tuple(flow.shape) != (B, D, H, W, 3):
            raise RuntimeError(f"warp: flow {tuple(flow.shape)} does not match src {tuple(src.shape)}")
        out = torch.empty_like(src)
        n = float(B) * D * H * W                # 4*(2C+3) B per voxel
        with _Guard(src, f"warp_fwd[C{C}]", n * (24.0 * C + 30.0), 4.0 * n * (2 * C + 3)):
            _lib.check(_L().modet_warp_fwd(_p(src), _p(flow), _p(out), B, D, H, W, C, mode, int(add_flow), _stream()),
                       "modet_warp_fwd")
        ctx.save_for_backward(src, flow)
        ctx.mode, ctx.add_flow, ctx.flow_bound = mode, int(add_flow), int(flow_bound)
        return out

    @staticmethod
    def backward(ctx, dout):
        src, flow = ctx.saved_tensors
        if ctx.mode != 0:
            raise RuntimeError("warp: nearest mode is not differentiable")
        dout = dout.contiguous()
        B, D, H, W, C = src.shape
        dsrc = torch.empty_like(src) if ctx.needs_input_grad[0] else None
        dflow = torch.empty_like(flow) if ctx.needs_input_grad[1] else None
        n = float(B) * D * H * W
        tag = _warp_bwd_tag(src, dsrc, ctx.add_flow, ctx.flow_bound if C == 3 else 0)     # the kernels that run
        with _Guard(src, tag, n * (60.0 * C + 40.0), 4.0 * n * (3 * C + 6)):
            _warp_backward(src, flow, dout, dsrc, dflow, None, ctx.add_flow, ctx.flow_bound if C == 3 else 0)
        return dsrc, dflow, None, None, None


# Deterministic warp backward (opt-in; modet_warp_bwd_det): d_src through 64-bit fixed-point integer atomics instead of float
# atomics -- bit-identical from run to run, and so is the whole train step (the warp scatter holds its only atomics).
DETERMINISTIC = os.environ.get("MODET_DETERMINISTIC", "0") == "1"


def set_deterministic(on=True):
    """run-to-run bit-identical gradients (costs ~0.3 ms per 160x192x160 train step: profiles/r05z_deterministic_fullsize.txt);
    returns the previous setting"""
    global DETERMINISTIC
    prev, DETERMINISTIC = DETERMINISTIC, bool(on)
    return prev


# The feature warps' backward by destination tiles (csrc/warp_tile.hip): integer sums in an LDS window instead of float atomics on
# global memory.  Round 5 built it (off: break-even in the step); round 6 rebuilt it (payload lists, zero-d_out entries dropped
# while binning, d_flow in the fill pass) and made it the default: faster AND bit-reproducible, so the plain step has no float
# atomics left and set_deterministic() costs nothing for these warps.  WARP_TILES = False brings the float-atomic kernel back
# (tools/ A/B runs).  Volumes below WARP_TILE_MIN_VOXELS keep it either way.
WARP_TILES = True
WARP_TILE_MIN_VOXELS = 0


def _warp_bwd_tag(src, dsrc, add_flow, flow_bound):
    """launch tag of a warp backward = the kernels that run (bench.py maps tags to kernel families); like _conv_tag only
    evaluated while a timer is installed"""
    if _TIMER is None:
        return None
    B, D, H, W, C = src.shape
    if C == 3 and flow_bound:
        return "warp_bwd_gather3[C3]"
    s16 = src.dtype == torch.bfloat16
    if (WARP_TILES and dsrc is not None and (C == 3 or not add_flow) and not (C == 3 and s16) and B * D * H * W >= WARP_TILE_MIN_VOXELS
            and _L().modet_warp_bwd_dsrc_tiles_ws_bytes(B, D, H, W, C)):
        return f"warp_bwd_tiles[C{C}]"
    return f"warp_bwd[C{C}]"


def _warp_backward(src, flow, dout, dsrc, dflow, galias, add_flow, flow_bound):
    """launch the warp backward: destination tiles (feature warps: C % 8 == 0, d_src wanted), else the float-atomic / gather
    kernel, else (deterministic mode, d_src of another channel count) 64-bit integer atomics on global memory.
    galias = a second flow gradient added on the way out"""
    B, D, H, W, C = src.shape
    L = _L()
    s16 = int(src.dtype == torch.bfloat16)
    if (WARP_TILES and dsrc is not None and not flow_bound and (C == 3 or not add_flow) and not (C == 3 and s16)
            and dout.dtype == torch.float32 and B * D * H * W >= WARP_TILE_MIN_VOXELS):
        nb = L.modet_warp_bwd_dsrc_tiles_ws_bytes(B, D, H, W, C)
        if nb:
            ws = _ws(nb, src)
            if dflow is not None:
                _lib.check(L.modet_warp_bwd_tiles(_p(src), s16, _p(flow), _p(dout), _p(dsrc), _p(dflow), _p(galias), _p(ws), nb,
                                                  B, D, H, W, C, int(add_flow), _stream()), "modet_warp_bwd_tiles")
            else:
                _lib.check(L.modet_warp_bwd_dsrc_tiles(_p(flow), _p(dout), _p(dsrc), _p(ws), nb, B, D, H, W, C, _stream()),
                           "modet_warp_bwd_dsrc_tiles")
            return
    if DETERMINISTIC and dsrc is not None and not flow_bound:
        nb = L.modet_warp_bwd_det_ws_bytes(B, D, H, W, C)
        ws = torch.empty((nb + 7) // 8, dtype=torch.int64, device=src.device)
        _lib.check(L.modet_warp_bwd_det(_p(src), s16, _p(flow), _p(dout), _p(dsrc), _p(dflow), _p(galias if dflow is not None else None),
                                        _p(ws), nb, B, D, H, W, C, int(add_flow), _stream()), "modet_warp_bwd_det")
        return
    _lib.check(L.modet_warp_bwd_acc(_p(src), s16, _p(flow), _p(dout), _p(dsrc), _p(dflow), _p(galias if dflow is not None else None),
                                    B, D, H, W, C, int(add_flow), int(flow_bound), _stream()), "modet_warp_bwd_acc")


class _WarpTee(Function):
    """warp(src, flow) for a flow that has a SECOND consumer: returns (out, flow_alias) -- the other consumer takes the alias.
    In the backward pass the other consumer's gradient arrives here (it ran later in the forward pass, so its backward ran
    first) and the warp kernel adds it while writing its own d_flow (modet_warp_bwd_acc): one pass instead of the
    element-wise add autograd would launch over two full-size tensors."""

    @staticmethod
    def forward(ctx, src, flow):
        _chk(src, flow)
        B, D, H, W, C = src.shape
        if tuple(flow.shape) != (B, D, H, W, 3):
            raise RuntimeError(f"warp: flow {tuple(flow.shape)} does not match src {tuple(src.shape)}")
        out = torch.empty_like(src)
        n = float(B) * D * H * W
        with _Guard(src, f"warp_fwd[C{C}]", n * (24.0 * C + 30.0), 4.0 * n * (2 * C + 3)):
            _lib.check(_L().modet_warp_fwd(_p(src), _p(flow), _p(out), B, D, H, W, C, 0, 0, _stream()), "modet_warp_fwd")
        ctx.save_for_backward(src, flow)
        ctx.set_materialize_grads(False)
        return out, flow.view_as(flow)

    @staticmethod
    def backward(ctx, dout, galias):
        src, flow = ctx.saved_tensors
        if dout is None:
            return None, galias
        dout = dout.contiguous()
        galias = None if galias is None else galias.contiguous()
        B, D, H, W, C = src.shape
        dsrc = torch.empty_like(src) if ctx.needs_input_grad[0] else None
        dflow = torch.empty_like(flow) if ctx.needs_input_grad[1] else None
        n = float(B) * D * H * W
        with _Guard(src, _warp_bwd_tag(src, dsrc, 0, 0), n * (60.0 * C + 40.0), 4.0 * n * (3 * C + 6 + (3 if galias is not None else 0))):
            _warp_backward(src, flow, dout, dsrc, dflow, galias, 0, 0)
        return dsrc, dflow


def warp_tee(src, flow):
    """(warp(src, flow), flow) for a flow with a second consumer, which must take the RETURNED flow; see _WarpTee"""
    return _WarpTee.apply(src, flow)


def cat_batch(a, b):
    """torch.cat([a, b], 0) -- without the copy when the two are adjacent halves of one buffer (engine.Trainer's static input
    buffer of a captured step: the [moving; fixed] batch of the shared encoder then costs nothing)"""
    if (a.shape == b.shape and a.is_contiguous() and b.is_contiguous() and a.dtype == b.dtype and a.device == b.device
            and a.untyped_storage().data_ptr() == b.untyped_storage().data_ptr()
            and b.storage_offset() == a.storage_offset() + a.numel() and not a.requires_grad and not b.requires_grad):
        return torch.as_strided(a, (2 * a.shape[0],) + tuple(a.shape[1:]), a.stride(), a.storage_offset())
    return torch.cat([a, b], 0)


def warp(src, flow, mode=0, add_flow=False, flow_bound=0):
    """SpatialTransformer on channels-last tensors; add_flow -> warp(src,flow)+flow.  reference: models.py:25-67.
    flow_bound=1 promises |flow| <= 1 voxel (attention outputs): the backward then gathers d_src without atomics."""
    return _Warp.apply(src, flow, mode, add_flow, flow_bound)


class _Upsample2(Function):
    @staticmethod
    def forward(ctx, x, scale):
        _chk(x)
        B, d, h, w, C = x.shape
        y = torch.empty((B, 2 * d, 2 * h, 2 * w, C), dtype=torch.float32, device=x.device)
        with _Guard(x, f"upsample2_fwd[C{C}]", 16.0 * y.numel(), 4.0 * (x.numel() + y.numel())):
            _lib.check(_L().modet_upsample2_fwd(_p(x), _p(y), B, d, h, w, C, float(scale), _stream()),
                       "modet_upsample2_fwd")
        ctx.shape, ctx.scale = (B, d, h, w, C), float(scale)
        return y

    @staticmethod
    def backward(ctx, dy):
        dy = dy.contiguous()
        B, d, h, w, C = ctx.shape
        dx = torch.empty(ctx.shape, dtype=torch.float32, device=dy.device)
        L = _L()
        nb = L.modet_upsample2_bwd_sep_ws_bytes(B, d, h, w, C)       # > 0: large level, three 1-D passes
        ws = _ws(nb, dy) if nb else None
        with _Guard(dy, f"upsample2_bwd[C{C}]", 16.0 * dy.numel(), 4.0 * (dx.numel() + dy.numel())):
            if nb:
                _lib.check(L.modet_upsample2_bwd_sep(_p(dy), _p(dx), _p(ws), nb, B, d, h, w, C, ctx.scale, _stream()),
                           "modet_upsample2_bwd_sep")
            else:
                _lib.check(L.modet_upsample2_bwd(_p(dy), _p(dx), B, d, h, w, C, ctx.scale, _stream()),
                           "modet_upsample2_bwd")
        return dx, None


def upsample2(x, scale=1.0):
    """scale * Upsample(x2, trilinear, align_corners=True), channels-last.  reference: models.py:354,:257-261"""
    return _Upsample2.apply(x, scale)


class _CwmTail(Function):
    @staticmethod
    def forward(ctx, x, logits):
        _chk(x, logits)
        heads = logits.shape[-1]
        N = logits.numel() // heads
        out = torch.empty(logits.shape[:-1] + (3,), dtype=torch.float32, device=x.device)
        with _Guard(x, "cwm_tail_fwd", 12.0 * x.numel(), 4.0 * (x.numel() + logits.numel() + out.numel())):
            _lib.check(_L().modet_cwm_tail_fwd(_p(x), _p(logits), _p(out), N, heads, _stream()), "modet_cwm_tail_fwd")
        ctx.save_for_backward(x, logits)
        return out

    @staticmethod
    def backward(ctx, dout):
        x, logits = ctx.saved_tensors
        dout = dout.contiguous()
        heads = logits.shape[-1]
        N = logits.numel() // heads
        dx, dl = torch.empty_like(x), torch.empty_like(logits)
        with _Guard(x, "cwm_tail_bwd", 20.0 * x.numel(), 8.0 * (x.numel() + logits.numel()) + 4.0 * dout.numel()):
            _lib.check(_L().modet_cwm_tail_bwd(_p(x), _p(logits), _p(dout), _p(dx), _p(dl), N, heads, _stream()),
                       "modet_cwm_tail_bwd")
        return dx, dl


def cwm_tail(x, logits):
    """2 * sum_h softmax_h(logits) * x[..., 3h:3h+3].  reference: CWM.forward, models.py:263-275"""
    return _CwmTail.apply(x, logits)


class _ToCL(Function):
    @staticmethod
    def forward(ctx, x):
        _chk(x)
        B, C = x.shape[0], x.shape[1]
        V = x.numel() // (B * C)
        if C == 1:
            return x.reshape((B,) + tuple(x.shape[2:]) + (1,))
        y = torch.empty((B,) + tuple(x.shape[2:]) + (C,), dtype=torch.float32, device=x.device)
        with _Guard(x, "layout", 0.0, 8.0 * x.numel()):
            _lib.check(_L().modet_ncdhw_to_cl(_p(x), _p(y), B, C, V, _stream()), "modet_ncdhw_to_cl")
        return y

    @staticmethod
    def backward(ctx, dy):
        return _ToNCDHW.apply(dy.contiguous())


class _ToNCDHW(Function):
    @staticmethod
    def forward(ctx, x):
        _chk(x)
        B, C = x.shape[0], x.shape[-1]
        V = x.numel() // (B * C)
        if C == 1:
            return x.reshape((B, 1) + tuple(x.shape[1:-1]))
        y = torch.empty((B, C) + tuple(x.shape[1:-1]), dtype=torch.float32, device=x.device)
        with _Guard(x, "layout", 0.0, 8.0 * x.numel()):
            _lib.check(_L().modet_cl_to_ncdhw(_p(x), _p(y), B, C, V, _stream()), "modet_cl_to_ncdhw")
        return y

    @staticmethod
    def backward(ctx, dy):
        return _ToCL.apply(dy.contiguous())


def to_channels_last(x):
    """(B,C,D,H,W) -> (B,D,H,W,C) (a free view when C == 1)"""
    return _ToCL.apply(x)


def to_ncdhw(x):
    """(B,D,H,W,C) -> (B,C,D,H,W)"""
    return _ToNCDHW.apply(x)


def _scale_by(x, s):
    y = torch.empty_like(x)
    with _Guard(x, "scale", x.numel(), 8.0 * x.numel()):
        _lib.check(_L().modet_scale_by_dev_scalar(_p(x), _p(s), _p(y), x.numel(), _stream()),
                   "modet_scale_by_dev_scalar")
    return y


def _ncc_launch(I, J, want_grad, win=9):
    """modet_ncc_fwd_bwd_win(I, J): (loss (1,), d loss / d J or None); win = 3 | 5 | 7 | 9 (cubic: the z-marching kernel) or a
    tuple (wz, wy, wx) of any positive sizes (modet_ncc_fwd_bwd_box: the reference's padding rule, separable sums)"""
    B, _, D, H, W = I.shape
    loss = torch.empty(1, dtype=torch.float32, device=I.device)
    dJ = torch.empty_like(J) if want_grad else None
    L = _L()
    if isinstance(win, tuple):
        wz, wy, wx = win
        nb = L.modet_ncc_box_ws_bytes(B, D, H, W, wz, wy, wx)
        if nb == 0:
            raise RuntimeError(f"NCC: window {list(win)} leaves no output voxel on a {(D, H, W)} volume")
        ws = _ws(nb, I)
        nv = float(I.numel())
        with _Guard(I, "ncc_fwd_bwd_box", 30.0 * (wz + wy + wx) * nv, 150.0 * nv):
            _lib.check(L.modet_ncc_fwd_bwd_box(_p(I), _p(J), _p(loss), _p(dJ), _p(ws), nb, B, D, H, W, wz, wy, wx, _stream()),
                       "modet_ncc_fwd_bwd_box")
        return loss, dJ
    nb = L.modet_ncc_ws_bytes(B, D, H, W)
    ws = _ws(nb, I)
    nv = float(I.numel())               # reads I,J once, writes d_J once
    with _Guard(I, "ncc_fwd_bwd", 400.0 * nv, 12.0 * nv):
        _lib.check(L.modet_ncc_fwd_bwd_win(_p(I), _p(J), _p(loss), _p(dJ), _p(ws), nb, B, D, H, W, int(win), _stream()),
                   "modet_ncc_fwd_bwd_win")
    return loss, dJ


class _NCC(Function):
    """cc is symmetric in its two arguments (losses.py:85-93), so the gradient w.r.t. y_true is the kernel's d_J with
    the roles swapped.  The reference's train loop differentiates the FIRST argument (train.py:127:
    ``loss_function(output[n], y)`` = NCC_vxm.forward(y_true=y_moved, y_pred=fixed))."""

    @staticmethod
    def forward(ctx, y_true, y_pred, win=9):
        _chk(y_true, y_pred)
        if y_true.shape != y_pred.shape or y_true.dim() != 5 or y_true.shape[1] != 1:
            raise RuntimeError("NCC: expects two (B,1,D,H,W) volumes")
        need_t, need_p = ctx.needs_input_grad[0], ctx.needs_input_grad[1]
        d_t = d_p = None
        if need_t and not need_p:
            loss, d_t = _ncc_launch(y_pred, y_true, True, win)
        else:
            loss, d_p = _ncc_launch(y_true, y_pred, need_p, win)
            if need_t:
                _, d_t = _ncc_launch(y_pred, y_true, True, win)
        ctx.save_for_backward(d_t, d_p)
        return loss.reshape(())

    @staticmethod
    def backward(ctx, g):
        d_t, d_p = ctx.saved_tensors
        g = g.contiguous().reshape(1)
        return (None if d_t is None else _scale_by(d_t, g)), (None if d_p is None else _scale_by(d_p, g)), None


def ncc_loss(y_true, y_pred, win=9):
    """NCC_vxm: -mean(cc) over the windows; win = an int (cubic) or [wz, wy, wx].  Cubic 3 / 5 / 7 / 9 run the z-marching
    kernel, every other window the general separable path.  reference: losses.py:34-94"""
    if isinstance(win, (list, tuple)):
        w = tuple(int(v) for v in win)
        if len(w) != 3 or min(w) < 1:
            raise RuntimeError(f"NCC: a window is three positive sizes, got {win}")
    else:
        w = (int(win),) * 3
    if w[0] == w[1] == w[2] and w[0] in (3, 5, 7, 9):
        return _NCC.apply(y_true, y_pred, w[0])
    return _NCC.apply(y_true, y_pred, w)


class _Grad3d(Function):
    @staticmethod
    def forward(ctx, flow, penalty):
        _chk(flow)
        if flow.dim() != 5 or flow.shape[1] != 3:
            raise RuntimeError("Grad3d: expects a (B,3,D,H,W) flow")
        B, _, D, H, W = flow.shape
        loss = torch.empty(1, dtype=torch.float32, device=flow.device)
        df = torch.empty_like(flow) if ctx.needs_input_grad[0] else None
        L = _L()
        nb = L.modet_grad3d_ws_bytes(B, D, H, W)
        ws = _ws(nb, flow)
        with _Guard(flow, "grad3d_fwd_bwd", 20.0 * flow.numel(), 8.0 * flow.numel()):
            _lib.check(L.modet_grad3d_fwd_bwd(_p(flow), _p(loss), _p(df), _p(ws), nb, B, D, H, W, int(penalty), _stream()),
                       "modet_grad3d_fwd_bwd")
        ctx.save_for_backward(df)
        return loss.reshape(())

    @staticmethod
    def backward(ctx, g):
        (df,) = ctx.saved_tensors
        return _scale_by(df, g.contiguous().reshape(1)), None


def grad3d_loss(flow, penalty="l2"):
    """Grad3d(penalty='l1' | 'l2') on a planar (B,3,D,H,W) flow.  reference: losses.py:6-31"""
    if penalty not in ("l1", "l2"):
        raise RuntimeError(f"Grad3d: unknown penalty {penalty!r}")
    return _Grad3d.apply(flow, 1 if penalty == "l1" else 2)


# ------------------------------------------------------------------------------------------------ non-autograd
def ncc_value_and_grad(y_true, y_pred, win=9, grad_scale=1.0):
    """(NCC_vxm(y_true, y_pred) as a device scalar, grad_scale * d loss / d y_pred) from one call -- no autograd node: the
    trainer seeds its backward with the gradient (engine.Trainer._seeded_loss).  Cubic windows 3 / 5 / 7 / 9.
    reference: losses.py:34-94, train.py:127"""
    _chk(y_true, y_pred)
    if y_true.shape != y_pred.shape or y_true.dim() != 5 or y_true.shape[1] != 1:
        raise RuntimeError("NCC: expects two (B,1,D,H,W) volumes")
    B, _, D, H, W = y_true.shape
    loss = torch.empty(1, dtype=torch.float32, device=y_true.device)
    dJ = torch.empty_like(y_pred)
    L = _L()
    nb = L.modet_ncc_ws_bytes(B, D, H, W)
    ws = _ws(nb, y_true)
    nv = float(y_true.numel())
    with _Guard(y_true, "ncc_fwd_bwd", 400.0 * nv, 12.0 * nv):
        _lib.check(L.modet_ncc_fwd_bwd_win_scaled(_p(y_true), _p(y_pred), _p(loss), _p(dJ), _p(ws), nb, B, D, H, W, int(win),
                                                  float(grad_scale), _stream()), "modet_ncc_fwd_bwd_win_scaled")
    return loss.reshape(()), dJ


def grad3d_value_and_grad_cl(flow_cl, penalty="l2", grad_scale=1.0):
    """(Grad3d(penalty) as a device scalar, grad_scale * d loss / d flow) of a CHANNELS-LAST flow (B,D,H,W,3), gradient in the
    same layout -- no autograd node, no planar copy of the flow.  reference: losses.py:6-31, train.py:128"""
    _chk(flow_cl)
    if flow_cl.dim() != 5 or flow_cl.shape[-1] != 3:
        raise RuntimeError("Grad3d: expects a channels-last (B,D,H,W,3) flow")
    if penalty not in ("l1", "l2"):
        raise RuntimeError(f"Grad3d: unknown penalty {penalty!r}")
    B, D, H, W, _ = flow_cl.shape
    loss = torch.empty(1, dtype=torch.float32, device=flow_cl.device)
    df = torch.empty_like(flow_cl)
    L = _L()
    nb = L.modet_grad3d_ws_bytes(B, D, H, W)
    ws = _ws(nb, flow_cl)
    with _Guard(flow_cl, "grad3d_fwd_bwd", 20.0 * flow_cl.numel(), 8.0 * flow_cl.numel()):
        _lib.check(L.modet_grad3d_fwd_bwd_cl(_p(flow_cl), _p(loss), _p(df), _p(ws), nb, B, D, H, W, 1 if penalty == "l1" else 2,
                                             float(grad_scale), _stream()), "modet_grad3d_fwd_bwd_cl")
    return loss.reshape(()), df


def adam_amsgrad_step_(p, g, m, v, vmax, lr, step, beta1=0.9, beta2=0.999, eps=1e-8, grad_scale=1.0):
    """in-place Adam(amsgrad=True) over flat buffers.  reference: train.py:101,:131-133"""
    _chk(p, g, m, v, vmax)
    with _Guard(p, "adam_amsgrad", 12.0 * p.numel(), 28.0 * p.numel()):
        _lib.check(_L().modet_adam_amsgrad_step(_p(p), _p(g), _p(m), _p(v), _p(vmax), p.numel(), float(lr), beta1,
                                                beta2, eps, int(step), float(grad_scale), _stream()),
                   "modet_adam_amsgrad_step")


def label_warp_counts(lab_moving, flow_cl, lab_fixed, nlabels=54, want_warped=True):
    """nearest label warp + Dice voxel counts on the GPU.  lab_* (D,H,W) int16, flow_cl (1,D,H,W,3).
    reference: utils.py:74-106, infer.py:86-92.  Returns (warped int16 or None, counts int64 (3, nlabels+1))."""
    if lab_moving.dtype != torch.int16 or lab_fixed.dtype != torch.int16:
        raise RuntimeError("label_warp_counts: labels must be int16")
    _chk(flow_cl)
    D, H, W = lab_moving.shape[-3:]
    lm, lf = lab_moving.contiguous(), lab_fixed.contiguous()
    warped = torch.empty((D, H, W), dtype=torch.int16, device=lm.device) if want_warped else None
    counts = torch.empty((3, nlabels + 1), dtype=torch.int64, device=lm.device)
    with _Guard(flow_cl):
        _lib.check(_L().modet_label_warp_counts(_p(lm), _p(flow_cl), _p(lf), _p(warped), _p(counts), D, H, W, nlabels,
                                                _stream()), "modet_label_warp_counts")
    return warped, counts


def jacdet_nonpos_count(flow_cl, want_det=False):
    """number of voxels per sample whose Jacobian determinant of (identity + flow) is <= 0, on the GPU, in the
    reference's fp64 operation order (utils.py:108-150, infer.py:89-90).  flow_cl (B,D,H,W,3) channels-last fp32.
    Returns (counts int64 (B,), det float64 (B,D,H,W) or None)."""
    _chk(flow_cl)
    B, D, H, W, C = flow_cl.shape
    if C != 3:
        raise RuntimeError("jacdet_nonpos_count: expects a (B,D,H,W,3) channels-last flow")
    counts = torch.empty(B, dtype=torch.int64, device=flow_cl.device)
    det = torch.empty((B, D, H, W), dtype=torch.float64, device=flow_cl.device) if want_det else None
    with _Guard(flow_cl, "jacdet", 60.0 * flow_cl.numel() / 3, 4.0 * flow_cl.numel()):
        _lib.check(_L().modet_jacdet_nonpos_count(_p(flow_cl), _p(counts), _p(det), B, D, H, W, _stream()),
                   "modet_jacdet_nonpos_count")
    return counts, det


# ------------------------------------------------------------------------------------------------ bf16 storage (cfg 5)
# BASELINE.json configs[4] "bf16 storage / fp32 accumulate": the ConvInsBlock chains (94 % of the step's FLOPs, half of its
# bytes) keep their activations in bf16 and run on the bf16 matrix pipe; everything that crosses a chain's boundary (level
# inputs / outputs, all parameters, statistics, weight gradients) stays fp32.  csrc/conv3d_bf16.hip, norm_act.hip.
def _chk16(*ts):
    for t in ts:
        if t is None:
            continue
        if not t.is_cuda:
            raise RuntimeError("smilecode_amd: tensor must live on the GPU (the HIP path has no CPU fallback)")
        if t.dtype not in (torch.float32, torch.bfloat16):
            raise RuntimeError(f"smilecode_amd: expected float32 or bfloat16, got {t.dtype}")
        if not t.is_contiguous():
            raise RuntimeError("smilecode_amd: tensor must be contiguous")


def _isbf(t):
    return 1 if t.dtype == torch.bfloat16 else 0


def cast_bf16(x, to_bf16):
    """fp32 <-> bf16 copy (round to nearest even) by the library's own kernel"""
    _chk16(x)
    y = torch.empty(x.shape, dtype=torch.bfloat16 if to_bf16 else torch.float32, device=x.device)
    with _Guard(x, "cast_bf16", 0.0, 6.0 * x.numel()):
        _lib.check(_L().modet_cast_bf16(_p(x), _p(y), x.numel(), int(to_bf16), _stream()), "modet_cast_bf16")
    return y


def conv3d_bf16_forward(x, w, b, want_stats=True, step=None):
    """y (bf16) = conv3d(x (fp32 | bf16), w) + b on the bf16 matrix pipe, fp32 accumulate; (y, stats | None)"""
    _chk16(x)
    _chk(w, b)
    step = step if step is not None else current_step()
    B, D, H, W, Cin = x.shape
    Cout = w.shape[0]
    if tuple(w.shape) != (Cout, Cin, 3, 3, 3):
        raise RuntimeError(f"conv3d_bf16: weight {tuple(w.shape)} does not match input channels {Cin}")
    L = _L()
    y = torch.empty((B, D, H, W, Cout), dtype=torch.bfloat16, device=x.device)
    nb = L.modet_conv3d_bf16_ws_bytes(Cin, Cout)
    ws = _ws(nb, x)
    sb = L.modet_conv3d_bf16_stats_bytes(B, D, H, W, Cin, Cout) if want_stats else 0
    stats = torch.empty(sb // 4, dtype=torch.float32, device=x.device) if sb > 0 else None
    n = float(B) * D * H * W
    with _Guard(x, _conv16_tag("fwd", x.shape, Cin, Cout, _isbf(x)), 54.0 * Cin * Cout * n, n * ((2.0 if _isbf(x) else 4.0) * Cin + 2.0 * Cout)):
        _lib.check(L.modet_conv3d_bf16_fwd(_p(x), _isbf(x), _p(w), _p(b), _p(y), _p(ws), nb, _p(stats), sb, B, D, H, W, Cin, Cout,
                                           _stream(), _h(step)), "modet_conv3d_bf16_fwd")
    return y, stats


def conv3d_bf16_backward_data(dy, w, Cin, dx_bf16, step=None):
    _chk16(dy)
    _chk(w)
    step = step if step is not None else current_step()
    B, D, H, W, Cout = dy.shape
    dx = torch.empty((B, D, H, W, Cin), dtype=torch.bfloat16 if dx_bf16 else torch.float32, device=dy.device)
    L = _L()
    nb = L.modet_conv3d_bf16_ws_bytes(Cin, Cout)
    ws = _ws(nb, dy)
    n = float(B) * D * H * W
    with _Guard(dy, _conv16_tag("dgrad", dy.shape, Cin, Cout, True), 54.0 * Cin * Cout * n, n * (2.0 * Cout + (2.0 if dx_bf16 else 4.0) * Cin)):
        _lib.check(L.modet_conv3d_bf16_bwd_data(_p(dy), _p(w), _p(dx), int(dx_bf16), _p(ws), nb, B, D, H, W, Cin, Cout, _stream(),
                                                _h(step)), "modet_conv3d_bf16_bwd_data")
    return dx


def conv3d_bf16_backward_weight(x, dy, w=None, b=None, step=None):
    """d_w, d_bias (fp32) from x (fp32 | bf16) and d_y (bf16); with a StepContext whose ``deferred()`` scope has
    destinations for the parameters ``w`` / ``b`` the gradients go there at the flush and (None, None) is returned"""
    _chk16(x, dy)
    L = _L()
    if hasattr(L, "modet_conv3d_bf16_bwd_weight"):
        B, D, H, W, Cin = x.shape
        Cout = dy.shape[-1]
        nb = L.modet_conv3d_bf16_bwd_weight_ws_bytes(B, D, H, W, Cin, Cout)
        ws = _ws(nb, x)
        n = float(B) * D * H * W
        scope = step if step is not None else current_step()
        dst = scope.destinations(w, b, True) if (scope is not None and hasattr(L, "modet_conv3d_bf16_bwd_weight_defer")) else None
        if dst is not None:
            dw, db = dst
            with _Guard(x, _conv16_tag("wgrad", x.shape, Cin, Cout, _isbf(x)), 54.0 * Cin * Cout * n, n * ((2.0 if _isbf(x) else 4.0) * Cin + 2.0 * Cout)):
                _lib.check(L.modet_conv3d_bf16_bwd_weight_defer(_p(x), _isbf(x), _p(dy), _p(dw), _p(db), _p(ws), nb, B, D, H, W,
                                                                Cin, Cout, _stream(), _h(scope)),
                           "modet_conv3d_bf16_bwd_weight_defer")
            scope._keep.append(ws)
            scope.written.add(w.data_ptr())
            scope.written.add(b.data_ptr())
            return None, None
        dw = torch.empty((Cout, Cin, 3, 3, 3), dtype=torch.float32, device=x.device)
        db = torch.empty((Cout,), dtype=torch.float32, device=x.device)
        with _Guard(x, _conv16_tag("wgrad", x.shape, Cin, Cout, _isbf(x)), 54.0 * Cin * Cout * n, n * ((2.0 if _isbf(x) else 4.0) * Cin + 2.0 * Cout)):
            _lib.check(L.modet_conv3d_bf16_bwd_weight(_p(x), _isbf(x), _p(dy), _p(dw), _p(db), _p(ws), nb, B, D, H, W, Cin, Cout,
                                                      _stream()), "modet_conv3d_bf16_bwd_weight")
        return dw, db
    x32 = x if x.dtype == torch.float32 else cast_bf16(x, False)
    return conv3d_backward_weight(x32, cast_bf16(dy, False), True)


class _Conv3dBF16(Function):
    """conv3d with bf16 output (+ InstanceNorm partial statistics from the fp32 accumulators, not differentiable)"""

    @staticmethod
    def forward(ctx, x, w, b):
        ctx.step = current_step()
        y, stats = conv3d_bf16_forward(x, w, b, True, ctx.step)
        ctx.save_for_backward(x, w, b)
        ctx.mark_non_differentiable(stats)
        ctx.set_materialize_grads(False)        # no zeros_like(stats) fill launch for the statistics output's "gradient"
        return y, stats

    @staticmethod
    def backward(ctx, dy, _dstats):
        if dy is None:
            return None, None, None
        x, w, b = ctx.saved_tensors
        dy = dy.contiguous()
        dx = conv3d_bf16_backward_data(dy, w, x.shape[-1], x.dtype == torch.bfloat16, ctx.step) if ctx.needs_input_grad[0] else None
        dw, db = conv3d_bf16_backward_weight(x, dy, w, b, ctx.step)
        return dx, dw, db


class _InstNormLReLUBF16(Function):
    """InstanceNorm3d + LeakyReLU(0.1) of a bf16 raw conv output; y is bf16 (inside a chain) or fp32 (a level's output)"""

    @staticmethod
    def forward(ctx, x, stats, eps, out_bf16):
        _chk16(x)
        if x.dtype != torch.bfloat16:
            raise RuntimeError("instnorm bf16: the raw conv output must be bfloat16")
        B, C = x.shape[0], x.shape[-1]
        V = x.numel() // (B * C)
        y = torch.empty(x.shape, dtype=torch.bfloat16 if out_bf16 else torch.float32, device=x.device)
        mean = torch.empty(B * C, dtype=torch.float32, device=x.device)
        rstd = torch.empty_like(mean)
        nel = float(x.numel())
        with _Guard(x, "instnorm_bf16_fwd", 8.0 * nel, nel * (4.0 if out_bf16 else 6.0)):
            _lib.check(_L().modet_instnorm_lrelu_fwd_stats_bf16(_p(x), _p(y), int(out_bf16), _p(mean), _p(rstd), _p(stats),
                                                                stats.numel() * 4, B, V, C, eps, _stream()),
                       "modet_instnorm_lrelu_fwd_stats_bf16")
        ctx.save_for_backward(x, mean, rstd)
        return y

    @staticmethod
    def backward(ctx, dy):
        x, mean, rstd = ctx.saved_tensors
        dy = dy.contiguous()
        B, C = x.shape[0], x.shape[-1]
        V = x.numel() // (B * C)
        dx = torch.empty_like(x)
        L = _L()
        nb = L.modet_instnorm_bf16_ws_bytes(B, V, C)
        ws = _ws(nb, x)
        nel = float(x.numel())
        with _Guard(x, "instnorm_bf16_bwd", 14.0 * nel, nel * (6.0 + 2.0 * (2.0 if _isbf(dy) else 4.0))):
            _lib.check(L.modet_instnorm_lrelu_bwd_bf16(_p(dy), _isbf(dy), _p(x), _p(mean), _p(rstd), _p(dx), _p(ws), nb, B, V, C,
                                                       _stream()), "modet_instnorm_lrelu_bwd_bf16")
        return dx, None, None, None


class _InstNormLReLUBF16PoolSplit(Function):
    """_InstNormLReLUBF16 (fp32 output) followed by _PoolTeeSplit as ONE node, bf16 chain: x_raw bf16 (2B,...) ->
    (avgpool2(y), y[:B], y[B:]) with y = LeakyReLU(InstanceNorm(x_raw)) in fp32.  Forward: the two kernels unchanged (a fused
    apply + pool was measured and bought nothing for the bf16 chain); backward: d_y = unpool(g_pooled) / 8 + [g_a ; g_b] is
    formed inside the two InstanceNorm backward passes (modet_instnorm_lrelu_bwd_pool_bf16) instead of written by the pool
    backward and read back twice: same arithmetic, same order of the sums -- bit-identical to the two-node form."""

    @staticmethod
    def forward(ctx, x, stats, eps, Bh, features16=False):
        _chk16(x)
        if x.dtype != torch.bfloat16:
            raise RuntimeError("instnorm bf16: the raw conv output must be bfloat16")
        B, D, H, W, C = x.shape
        V = D * H * W
        L = _L()
        y = torch.empty(x.shape, dtype=torch.bfloat16 if features16 else torch.float32, device=x.device)
        mean = torch.empty(B * C, dtype=torch.float32, device=x.device)
        rstd = torch.empty_like(mean)
        pooled = torch.empty((B, D // 2, H // 2, W // 2, C), dtype=torch.float32, device=x.device)
        nel = float(x.numel())
        if features16:
            # one pass: bf16 features + the pooled tensor from their fp32 values (pooling the ROUNDED features costs the gradient)
            with _Guard(x, "instnorm_bf16_fwd", 9.0 * nel, nel * 4.5):
                _lib.check(L.modet_instnorm_lrelu_fwd_stats_pool_bf16(_p(x), _p(y), _p(pooled), _p(mean), _p(rstd), _p(stats),
                                                                      stats.numel() * 4, B, D, H, W, C, eps, _stream()),
                           "modet_instnorm_lrelu_fwd_stats_pool_bf16")
        else:
            with _Guard(x, "instnorm_bf16_fwd", 8.0 * nel, nel * 6.0):
                _lib.check(L.modet_instnorm_lrelu_fwd_stats_bf16(_p(x), _p(y), 0, _p(mean), _p(rstd), _p(stats), stats.numel() * 4, B,
                                                                 V, C, eps, _stream()), "modet_instnorm_lrelu_fwd_stats_bf16")
            with _Guard(y, "avgpool2_fwd", y.numel(), 4.5 * y.numel()):
                _lib.check(L.modet_avgpool2_fwd(_p(y), _p(pooled), B, D, H, W, C, _stream()), "modet_avgpool2_fwd")
        ctx.save_for_backward(x, mean, rstd)
        ctx.Bh = Bh
        ctx.set_materialize_grads(False)
        if not features16:
            return pooled, y[:Bh], y[Bh:]
        # bf16 level features: the two halves leave as fp32 HANDLES -- allocated, never written, never read -- that carry the
        # bf16 tensor as ``.data16``.  Autograd sees fp32 tensors of the right shape, so the consumers' fp32 gradients arrive
        # here uncast (a bf16 tensor on the edge would have them rounded to bf16, with a cast pass each); the consumers
        # (level_attention_bf16) read ``.data16`` only.
        # (one element of storage expanded to the shape -- autograd checks sizes only -- not a full fp32 allocation
        # beside the bf16 data; and only ModeT.forward ever asks for handles: Encoder.forward_pair(handles=True))
        hm = torch.empty(1, dtype=torch.float32, device=x.device).expand((Bh,) + tuple(x.shape[1:]))
        hf = torch.empty(1, dtype=torch.float32, device=x.device).expand((B - Bh,) + tuple(x.shape[1:]))
        hm.data16, hf.data16 = y[:Bh], y[Bh:]
        return pooled, hm, hf

    @staticmethod
    def backward(ctx, gy, ga, gb):
        if not ctx.needs_input_grad[0] or (gy is None and ga is None and gb is None):
            return None, None, None, None, None
        x, mean, rstd = ctx.saved_tensors
        B, D, H, W, C = x.shape
        Bh = ctx.Bh
        V = D * H * W
        L = _L()
        dx = torch.empty_like(x)
        nb = L.modet_instnorm_bf16_ws_bytes(B, V, C)
        ws = _ws(nb, x)
        nel = float(x.numel())
        if gy is not None:
            gy = gy.contiguous()
            ga = None if ga is None else ga.contiguous()
            gb = None if gb is None else gb.contiguous()
            with _Guard(x, "instnorm_bf16_bwd", 15.0 * nel, nel * 14.5):
                _lib.check(L.modet_instnorm_lrelu_bwd_pool_bf16(_p(gy), _p(ga), _p(gb), Bh, _p(x), _p(mean), _p(rstd), _p(dx), _p(ws),
                                                                nb, B, D, H, W, C, _stream()), "modet_instnorm_lrelu_bwd_pool_bf16")
            return dx, None, None, None, None
        dy = torch.empty(x.shape, dtype=torch.float32, device=x.device)
        for sl, g in ((slice(0, Bh), ga), (slice(Bh, B), gb)):
            if g is None:
                dy[sl].zero_()
            else:
                dy[sl].copy_(g)
        with _Guard(x, "instnorm_bf16_bwd", 14.0 * nel, nel * 14.0):
            _lib.check(L.modet_instnorm_lrelu_bwd_bf16(_p(dy), 0, _p(x), _p(mean), _p(rstd), _p(dx), _p(ws), nb, B, V, C, _stream()),
                       "modet_instnorm_lrelu_bwd_bf16")
        return dx, None, None, None, None


def conv_ins_pair_bf16_pool_split(inp, w1, b1, w2, b2, Bh, eps=1e-5, features16=False):
    """conv_ins_pair_bf16 whose output goes to AvgPool3d(2) and, split into its two batch halves, to the level's consumers:
    (pooled, y[:Bh], y[Bh:]); see _InstNormLReLUBF16PoolSplit.  features16: the two halves are stored as bf16 and handed out as
    fp32 handles with ``.data16`` (only level_attention_bf16 understands those)."""
    raw1, st1 = _Conv3dBF16.apply(inp, w1, b1)
    y1 = _InstNormLReLUBF16.apply(raw1, st1, eps, True)
    raw2, st2 = _Conv3dBF16.apply(y1, w2, b2)
    return _InstNormLReLUBF16PoolSplit.apply(raw2, st2, eps, Bh, features16)


def feature_handle_like(new, old):
    """``new`` (a detached copy / leaf of the fp32 handle ``old``) carries the same bf16 level features"""
    d16 = getattr(old, "data16", None)
    if d16 is not None:
        new.data16 = d16
    return new


def conv_ins_pair_bf16(inp, w1, b1, w2, b2, eps=1e-5):
    """ConvInsBlock -> ConvInsBlock with bf16 storage inside the chain: fp32 (or bf16) in, fp32 out"""
    raw1, st1 = _Conv3dBF16.apply(inp, w1, b1)
    y1 = _InstNormLReLUBF16.apply(raw1, st1, eps, True)
    raw2, st2 = _Conv3dBF16.apply(y1, w2, b2)
    return _InstNormLReLUBF16.apply(raw2, st2, eps, False)
