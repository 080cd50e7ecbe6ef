"""The feature warps' backward on the REAL train step's tensors (160x192x160, synthetic pair 24, the bench's weights): every
_warp_backward call of one Trainer._fwd_bwd is recorded (src, flow, d_out, second flow gradient), then

    python tools/warp_real.py op        # per call: float-atomic kernel vs destination tiles, HIP-event ms + max differences
    python tools/warp_real.py tiles N   # N runs of the tile path on the level-1 call only (for rocprofv3 --kernel-trace --stats)
    python tools/warp_real.py step      # same-process A/B of the captured train step, ops.WARP_TILES on / off
"""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from smilecode_amd import models, ops, synth  # noqa: E402
from smilecode_amd.engine import Trainer  # noqa: E402

shape = tuple(int(v) for v in os.environ.get("SHAPE", "160,192,160").split(","))


def setup():
    model = models.ModeT(shape, head_dim=6, num_heads=[8, 4, 2, 1, 1], scale=1).cuda()
    models.load_numpy_weights(model, synth.make_weights(24))
    mov, fix = (torch.from_numpy(a).cuda() for a in synth.make_pair(shape, 24))
    return model, mov, fix


def record(model, mov, fix):
    calls = []
    orig = ops._warp_backward

    def spy(src, flow, dout, dsrc, dflow, galias, add_flow, flow_bound):
        if dsrc is not None and not flow_bound and not add_flow:
            calls.append(tuple(None if t is None else t.detach().clone() for t in (src, flow, dout, galias)) + (dflow is not None,))
        return orig(src, flow, dout, dsrc, dflow, galias, add_flow, flow_bound)
    ops._warp_backward = spy
    tr = Trainer(model)
    tr._fwd_bwd(mov, fix)
    torch.cuda.synchronize()
    ops._warp_backward = orig
    return calls


def timed(fn, n=20):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        fn()
    e1.record()
    e1.synchronize()
    return e0.elapsed_time(e1) / n


def run(call, tiles):
    src, flow, dout, galias, want_flow = call
    dsrc = torch.empty_like(src, dtype=torch.float32)
    dflow = torch.empty_like(flow) if want_flow else None

    def fn():
        ops.WARP_TILES = tiles
        ops._warp_backward(src, flow, dout, dsrc, dflow, galias, 0, 0)
    return fn, dsrc, dflow


def main():
    mode = sys.argv[1] if len(sys.argv) > 1 else "op"
    model, mov, fix = setup()
    if mode == "step":
        variants = {}
        for v, mv in ((False, 0), (True, 0), (True, 400_000)):
            ops.WARP_TILES, ops.WARP_TILE_MIN_VOXELS = v, mv
            tr = Trainer(model)
            tr.capture(mov, fix)
            variants[f"WARP_TILES={v} min_voxels={mv}"] = tr
        ops.WARP_TILES, ops.WARP_TILE_MIN_VOXELS = True, 0
        for tr in variants.values():
            for _ in range(3):
                tr._graph.replay()
        torch.cuda.synchronize()
        res = {k: [] for k in variants}
        for rep in range(8):
            for name, tr in variants.items():
                torch.cuda.synchronize()
                t0 = time.perf_counter()
                for _ in range(20):
                    tr._graph.replay()
                torch.cuda.synchronize()
                res[name].append((time.perf_counter() - t0) / 20 * 1e3)
        for name, v in res.items():
            v.sort()
            print("%-50s median %.3f ms   min %.3f   max %.3f" % (name, v[len(v) // 2], v[0], v[-1]))
        return
    calls = record(model, mov, fix)
    if mode == "tiles":
        n = int(sys.argv[2]) if len(sys.argv) > 2 else 10
        only = int(os.environ.get("CALL", "-1"))
        for i, c in enumerate(calls):
            if only >= 0 and i != only:
                continue
            if os.environ.get("DSRC_ONLY"):
                c = c[:4] + (False,)
            fn, _, _ = run(c, True)
            for _ in range(n):
                fn()
        torch.cuda.synchronize()
        return
    for c in calls:
        src, flow, dout, galias, want_flow = c
        B, D, H, W, C = src.shape
        nz = float((dout.abs().amax(dim=-1) > 0).float().mean())
        fa, a_s, a_f = run(c, False)
        ta = timed(fa)
        ft, t_s, t_f = run(c, True)
        tt = timed(ft)
        fa(); ft()
        torch.cuda.synchronize()
        es = float((a_s - t_s).abs().max()) / max(float(a_s.abs().max()), 1e-30)
        ef = float((a_f - t_f).abs().max()) / max(float(a_f.abs().max()), 1e-30) if want_flow else float("nan")
        print(f"C={C:3d} {D}x{H}x{W} src {str(src.dtype)[6:]:8s} d_out non-zero voxels {nz:.3f} |flow|max {float(flow.abs().max()):.2f}: "
              f"atomics {ta:.4f} ms   tiles {tt:.4f} ms   d_src diff {es:.2e} of max, d_flow diff {ef:.2e} of max", flush=True)


if __name__ == "__main__":
    main()
