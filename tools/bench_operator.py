"""The operator boundary on its own: ``modetqkrpb_cu`` forward and backward (smilecode_amd/functional.py ->
modet_qk_fwd / modet_qk_bwd) at the five shapes a ModeT-cu step calls it with (ModeT-cu/models.py:323-352: head_dim 6,
heads 8,4,2,1,1 from the coarsest level to the finest; LPBA 160x192x160, batch 1).

    python tools/bench_operator.py [--iters 20] [--dtype f32|f64] [--json out.json]

Algorithmic bytes per voxel and head (E = element size):  forward  q 6E + kpad 6E (+ring) + attn 27E;
backward  d_attn 27E + q 6E + kpad 6E + d_q 6E + d_kpad 6E.  HIP events on the launch stream, median over --iters."""
import argparse
import json
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from smilecode_amd.functional import modet_bw, modet_fw  # noqa: E402

LEVELS = [((160, 192, 160), 1), ((80, 96, 80), 1), ((40, 48, 40), 2), ((20, 24, 20), 4), ((10, 12, 10), 8)]


def timed(fn, iters):
    for _ in range(3):
        fn()
    ts = []
    for _ in range(iters):
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record(); fn(); b.record(); b.synchronize()
        ts.append(a.elapsed_time(b))
    ts.sort()
    return ts[len(ts) // 2]


def run(iters=20, dtype=torch.float32, levels=LEVELS, hd=6):
    E = torch.empty((), dtype=dtype).element_size()
    rows = []
    for (D, H, W), heads in levels:
        g = torch.Generator(device="cuda").manual_seed(D)
        V, Vp = D * H * W, (D + 2) * (H + 2) * (W + 2)
        q = torch.randn((1, heads, D, H, W, hd), device="cuda", dtype=dtype, generator=g)
        k = torch.zeros((1, heads, D + 2, H + 2, W + 2, hd), device="cuda", dtype=dtype)
        k[:, :, 1:-1, 1:-1, 1:-1] = torch.randn((1, heads, D, H, W, hd), device="cuda", dtype=dtype, generator=g)
        rpb = torch.randn((heads, 3, 3, 3), device="cuda", dtype=dtype, generator=g)
        ga = torch.randn((1, heads, D, H, W, 27), device="cuda", dtype=dtype, generator=g)
        t_f = timed(lambda: modet_fw(q, k, rpb), iters)
        t_b = timed(lambda: modet_bw(ga, q, k, True), iters)
        by_f = heads * E * (V * (hd + 27) + Vp * hd)
        by_b = heads * E * (V * (27 + 2 * hd) + 2 * Vp * hd)
        rows.append({"shape": [D, H, W], "heads": heads, "fwd_ms": t_f, "bwd_ms": t_b, "fwd_bytes": by_f, "bwd_bytes": by_b,
                     "fwd_GBps": by_f / t_f / 1e6, "bwd_GBps": by_b / t_b / 1e6})
        del q, k, ga
        torch.cuda.empty_cache()
    return rows


if __name__ == "__main__":
    ap = argparse.ArgumentParser()
    ap.add_argument("--iters", type=int, default=20)
    ap.add_argument("--dtype", choices=["f32", "f64"], default="f32")
    ap.add_argument("--json", default=None)
    ap.add_argument("--level", type=int, default=0, help="1..5: only that level (for rocprofv3 runs); 0 = all five")
    a = ap.parse_args()
    rows = run(a.iters, torch.float32 if a.dtype == "f32" else torch.float64,
               LEVELS if a.level == 0 else LEVELS[a.level - 1:a.level])
    for r in rows:
        print("%-16s heads %d  fwd %.3f ms (%.0f GB/s)  bwd %.3f ms (%.0f GB/s)" % (
            "x".join(map(str, r["shape"])), r["heads"], r["fwd_ms"], r["fwd_GBps"], r["bwd_ms"], r["bwd_GBps"]))
    tot = sum(r["fwd_ms"] + r["bwd_ms"] for r in rows)
    print("sum over the five levels: %.3f ms" % tot)
    if a.json:
        json.dump({"dtype": a.dtype, "levels": rows, "sum_ms": tot}, open(a.json, "w"), indent=1)
