"""Per-stage error attribution of the HIP forward against the fp64 CPU oracle (VERDICT r1 "What's weak" 1).

For every stage of ModeT.forward two numbers are reported, both against the fp64 oracle's tap of that stage:
  * ``acc``   -- the HIP pipeline's own tensor (error accumulated from the inputs up to here);
  * ``local`` -- the HIP stage fed with the ORACLE's (fp64 -> fp32) inputs of that stage, i.e. the error this stage adds;
and the same two for the ATen fp32 CPU pipeline (``cpu32``), which is the yardstick: a HIP stage whose local error is far
above ATen's is the one to fix.  Writes profiles/<out>.json.

    python tools/attrib_fullsize.py [--shape 160,192,160] [--out r02_attrib_fullsize]
"""
import argparse
import json
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

from oracle import modet_torch as orc                      # noqa: E402  (checker)
from smilecode_amd import models, ops, synth               # noqa: E402

HEADS = (8, 4, 2, 1, 1)


def cl(t):
    """oracle NCDHW (cpu, any float) -> cuda fp32 channels-last"""
    return t.float().permute(0, 2, 3, 4, 1).contiguous().cuda()


def back(t_cl):
    """cuda channels-last -> cpu fp64 NCDHW"""
    return t_cl.detach().permute(0, 4, 1, 2, 3).double().cpu()


def err(got, ref):
    d = (got.double() - ref.double())
    return {"max": float(d.abs().max()), "rms": float(d.pow(2).mean().sqrt()), "ref_absmax": float(ref.abs().max()),
            "ref_rms": float(ref.double().pow(2).mean().sqrt())}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--shape", default="160,192,160")
    ap.add_argument("--out", default="r02_attrib_fullsize")
    ap.add_argument("--seed", type=int, default=24)
    args = ap.parse_args()
    shape = tuple(int(s) for s in args.shape.split(","))
    w = synth.make_weights(24)
    mov_np, fix_np = synth.make_pair(shape, args.seed)
    mov, fix = torch.from_numpy(mov_np), torch.from_numpy(fix_np)

    t0 = time.time()
    p64 = {n: torch.from_numpy(v).double() for n, v in w.items()}
    T64 = {}
    with torch.no_grad():
        y64, f64 = orc.modet_forward(p64, mov.double(), fix.double(), HEADS, 6, 1.0, taps=T64)
    T64["flow1"], T64["y"] = f64, y64
    print(f"[attrib] fp64 oracle {time.time() - t0:.1f}s", flush=True)
    t0 = time.time()
    p32 = {n: torch.from_numpy(v) for n, v in w.items()}
    T32 = {}
    with torch.no_grad():
        y32, f32 = orc.modet_forward(p32, mov, fix, HEADS, 6, 1.0, taps=T32)
    T32["flow1"], T32["y"] = f32, y32
    print(f"[attrib] fp32 oracle {time.time() - t0:.1f}s", flush=True)

    model = models.ModeT(shape, head_dim=6, num_heads=list(HEADS), scale=1).cuda().eval()
    models.load_numpy_weights(model, w)
    rep = {"shape": list(shape), "stages": {}}

    def note(name, kind, got, ref):
        rep["stages"].setdefault(name, {})[kind] = err(got, ref)

    def cpu32_local(name, fn):
        """ATen fp32 on the fp64 oracle's inputs of the stage"""
        with torch.no_grad():
            note(name, "cpu32_local", fn(), T64[name])

    with torch.no_grad():
        # ---------------------------------------------------------------- accumulated: the HIP pipeline itself, stage by stage
        mov_cl, fix_cl = ops.to_channels_last(mov.cuda()), ops.to_channels_last(fix.cuda())
        M, Fx = model.encoder.forward_pair(torch.cat([mov_cl, fix_cl], 0), 1)
        A = {}
        for i in range(5):
            A[f"M{i + 1}"], A[f"F{i + 1}"] = M[i], Fx[i]
        ST = model.transformer

        def level(lvl, Ff, Mf):
            q = getattr(model, f"projblock{lvl}")(Ff)
            k = getattr(model, f"projblock{lvl}")(Mf)
            m = getattr(model, f"mdt{lvl}")(q, k)
            A[f"q{lvl}"], A[f"k{lvl}"], A[f"mdt{lvl}"] = q, k, m
            wv = getattr(model, f"cwm{lvl}")(m) if lvl >= 3 else m
            A[f"w{lvl}"] = wv
            return wv

        flow = level(5, Fx[4], M[4]); A["flow5"] = flow
        A["Mw4"] = ST[3].forward_cl(M[3], flow)
        wv = level(4, Fx[3], A["Mw4"])
        flow = ST[2].forward_cl(ops.upsample2(flow, 2.0), wv, add_flow=True); A["flow4"] = flow
        A["Mw3"] = ST[2].forward_cl(M[2], flow)
        wv = level(3, Fx[2], A["Mw3"])
        flow = ST[1].forward_cl(ops.upsample2(flow, 2.0), wv, add_flow=True); A["flow3"] = flow
        A["Mw2"] = ST[1].forward_cl(M[1], flow)
        wv = level(2, Fx[1], A["Mw2"])
        flow = ops.upsample2(ST[1].forward_cl(flow, wv, add_flow=True, flow_bound=1), 2.0); A["flow2"] = flow
        A["Mw1"] = ST[0].forward_cl(M[0], flow)
        wv = level(1, Fx[0], A["Mw1"])
        flow = ST[0].forward_cl(flow, wv, add_flow=True, flow_bound=1); A["flow1"] = flow
        A["y"] = ST[0].forward_cl(mov_cl, flow)
        for name, t in A.items():
            ref = T64[name]
            got = back(t)
            if name[0] in "qk":                       # oracle q/k are channels-last already
                got = t.detach().double().cpu()
            note(name, "acc", got, ref)
            note(name, "cpu32_acc", T32[name], ref)
        # cross-check against the module's own forward
        y_mod, f_mod = model(mov.cuda(), fix.cuda())
        assert torch.equal(ops.to_ncdhw(A["flow1"]), f_mod), "staged forward differs from ModeT.forward"
        del A
        torch.cuda.empty_cache()

        # ---------------------------------------------------------------- local: each HIP stage on the oracle's inputs
        enc = model.encoder
        for tag, img in (("M", mov), ("F", fix)):
            x = ops.to_channels_last(img.cuda())
            note(f"enc{tag}.0.0", "local", back(enc.conv0[0](x)), T64[f"enc{tag}.0.0"])
            h = models._two_blocks(cl(T64[f"enc{tag}.0.0"]), enc.conv0[1], enc.conv0[2])
            note(f"enc{tag}.0.2", "local", back(h), T64[f"enc{tag}.0.2"])
            h1 = ops.conv3d_instnorm_lrelu(cl(T64[f"enc{tag}.0.0"]), enc.conv0[1].main.weight, enc.conv0[1].main.bias)
            note(f"enc{tag}.0.1", "local", back(h1), T64[f"enc{tag}.0.1"])
            h2 = ops.conv3d_instnorm_lrelu(cl(T64[f"enc{tag}.0.1"]), enc.conv0[2].main.weight, enc.conv0[2].main.bias)
            note(f"enc{tag}.0.2", "local_single", back(h2), T64[f"enc{tag}.0.2"])
            for lvl, blk in zip(range(1, 5), (enc.conv1, enc.conv2, enc.conv3, enc.conv4)):
                pooled = ops.avgpool2(cl(T64[f"enc{tag}.{lvl - 1}.2"]))
                h = models._two_blocks(pooled, blk[1], blk[2])
                note(f"enc{tag}.{lvl}.2", "local", back(h), T64[f"enc{tag}.{lvl}.2"])
            # ATen fp32 on the same oracle inputs
            for lvl in range(0, 5):
                src = T64[f"enc{tag}.0.0"].float() if lvl == 0 else torch.nn.functional.avg_pool3d(T64[f"enc{tag}.{lvl - 1}.2"].float(), 2)
                a = "0.1" if lvl == 0 else f"{lvl}.1"
                b = "0.2" if lvl == 0 else f"{lvl}.2"
                h = orc.conv_ins_block(p32, f"encoder.conv{lvl}.{a[-1]}", src)
                h = orc.conv_ins_block(p32, f"encoder.conv{lvl}.{b[-1]}", h)
                note(f"enc{tag}.{lvl}.2", "cpu32_local", h, T64[f"enc{tag}.{lvl}.2"])
        torch.cuda.empty_cache()

        flow_in = {4: "flow5", 3: "flow4", 2: "flow3", 1: "flow2"}
        for lvl in (5, 4, 3, 2, 1):
            heads = HEADS[5 - lvl]
            pb, mdt = getattr(model, f"projblock{lvl}"), getattr(model, f"mdt{lvl}")
            if lvl < 5:
                note(f"Mw{lvl}", "local", back(ST[lvl - 1].forward_cl(cl(T64[f"M{lvl}"]), cl(T64[flow_in[lvl]]))), T64[f"Mw{lvl}"])
                note(f"Mw{lvl}", "cpu32_local", orc.warp(T64[f"M{lvl}"].float(), T64[flow_in[lvl]].float()), T64[f"Mw{lvl}"])
                kin = T64[f"Mw{lvl}"]
            else:
                kin = T64["M5"]
            note(f"q{lvl}", "local", pb(cl(T64[f"F{lvl}"])).double().cpu(), T64[f"q{lvl}"])
            note(f"k{lvl}", "local", pb(cl(kin)).double().cpu(), T64[f"k{lvl}"])
            note(f"q{lvl}", "cpu32_local", orc.projection(p32, f"projblock{lvl}", T64[f"F{lvl}"].float()), T64[f"q{lvl}"])
            note(f"k{lvl}", "cpu32_local", orc.projection(p32, f"projblock{lvl}", kin.float()), T64[f"k{lvl}"])
            q32, k32 = T64[f"q{lvl}"].float(), T64[f"k{lvl}"].float()
            note(f"mdt{lvl}", "local", back(mdt(q32.cuda().contiguous(), k32.cuda().contiguous())), T64[f"mdt{lvl}"])
            note(f"mdt{lvl}", "cpu32_local", orc.mode_transformer(q32, k32, p32[f"mdt{lvl}.rpb"], heads, 1.0), T64[f"mdt{lvl}"])
            if lvl >= 3:
                note(f"w{lvl}", "local", back(getattr(model, f"cwm{lvl}")(cl(T64[f"mdt{lvl}"]))), T64[f"w{lvl}"])
                note(f"w{lvl}", "cpu32_local", orc.cwm(p32, f"cwm{lvl}", T64[f"mdt{lvl}"].float(), heads), T64[f"w{lvl}"])
            torch.cuda.empty_cache()
        # compositions
        for lvl, fin in ((4, "flow5"), (3, "flow4")):
            got = ST[lvl - 2].forward_cl(ops.upsample2(cl(T64[fin]), 2.0), cl(T64[f"w{lvl}"]), add_flow=True)
            note(f"flow{lvl}", "local", back(got), T64[f"flow{lvl}"])
            note(f"flow{lvl}", "cpu32_local", orc.warp(orc.upsample2(2 * T64[fin].float()), T64[f"w{lvl}"].float()) + T64[f"w{lvl}"].float(),
                 T64[f"flow{lvl}"])
        got = ops.upsample2(ST[1].forward_cl(cl(T64["flow3"]), cl(T64["w2"]), add_flow=True, flow_bound=1), 2.0)
        note("flow2", "local", back(got), T64["flow2"])
        note("flow2", "cpu32_local", orc.upsample2(2 * (orc.warp(T64["flow3"].float(), T64["w2"].float()) + T64["w2"].float())), T64["flow2"])
        got = ST[0].forward_cl(cl(T64["flow2"]), cl(T64["w1"]), add_flow=True, flow_bound=1)
        note("flow1", "local", back(got), T64["flow1"])
        note("flow1", "cpu32_local", orc.warp(T64["flow2"].float(), T64["w1"].float()) + T64["w1"].float(), T64["flow1"])
        note("y", "local", back(ST[0].forward_cl(mov_cl, cl(T64["flow1"]))), T64["y"])
        note("y", "cpu32_local", orc.warp(mov, T64["flow1"].float()), T64["y"])

    os.makedirs(os.path.join(ROOT, "profiles"), exist_ok=True)
    out = os.path.join(ROOT, "profiles", args.out + ".json")
    with open(out, "w") as f:
        json.dump(rep, f, indent=1, sort_keys=True)
    os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
    with open(os.path.join(ROOT, "gpurun_out", args.out + ".json"), "w") as f:
        json.dump(rep, f, indent=1, sort_keys=True)
    order = ([f"enc{t}.{s}" for t in "MF" for s in ("0.0", "0.1", "0.2", "1.2", "2.2", "3.2", "4.2")]
             + [f"{a}{l}" for l in (5, 4, 3, 2, 1) for a in ("Mw", "q", "k", "mdt", "w", "flow")] + ["y"])
    print(f"{'stage':10s} {'ref_max':>9s} | {'hip acc':>9s} {'cpu32 acc':>9s} | {'hip loc':>9s} {'cpu32 loc':>9s}   (max abs err)")
    for name in order:
        s = rep["stages"].get(name)
        if not s:
            continue
        g = lambda k: f"{s[k]['max']:9.2e}" if k in s else "        -"
        any_ = next(iter(s.values()))
        print(f"{name:10s} {any_['ref_absmax']:9.2e} | {g('acc')} {g('cpu32_acc')} | {g('local')} {g('cpu32_local')}"
              + (f"  single {s['local_single']['max']:.2e}" if "local_single" in s else ""))


if __name__ == "__main__":
    main()
