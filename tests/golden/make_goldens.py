"""Generate the committed golden vectors from the REAL reference (build container only).

    python tests/golden/make_goldens.py

Imports /root/reference/ModeT/{models,losses}.py read-only (never copied, never shipped:
/root/reference does not exist on the GPU box), runs it on CPU in fp64 on seeded inputs
from smilecode_amd.synth, and writes small .npz fixtures next to this file.  While doing so it
checks oracle/modet_torch.py (our functional restatement) against the reference and writes
the max deviations to tests/golden/REPORT.txt -- that is the oracle's parity pin.

Fixtures are data only: inputs (or their synth seeds) and the reference's outputs.
"""
import os
import sys
import warnings

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
sys.path.insert(0, "/root/reference/ModeT")
warnings.filterwarnings("ignore")

import models as ref_models  # noqa: E402  (reference)
import losses as ref_losses  # noqa: E402  (reference)

from smilecode_amd import synth  # noqa: E402
from oracle import modet_torch as orc  # noqa: E402

torch.set_num_threads(8)
REPORT = []
HEADS = [8, 4, 2, 1, 1]


def report(name, a, b):
    err = float((a.detach().double() - b.detach().double()).abs().max())
    mag = float(b.detach().double().abs().max())
    REPORT.append(f"{name:58s} max|oracle-ref| = {err:.3e}   max|ref| = {mag:.3e}")
    return err


def T(a, dt=torch.float64):
    return torch.from_numpy(np.ascontiguousarray(a)).to(dt)


def ref_model(shape, scale, dt=torch.float64, seed=24):
    m = ref_models.ModeT(shape, head_dim=6, num_heads=HEADS, scale=scale)
    w = synth.make_weights(seed)
    spec = synth.param_spec()
    assert [n for n, _ in m.named_parameters()] == list(spec.keys()), "param_spec drifted from the reference"
    for n, p in m.named_parameters():
        assert tuple(p.shape) == spec[n], n
    sd = m.state_dict()
    for n, v in w.items():
        sd[n] = T(v, torch.float32)
    m.load_state_dict(sd)
    return m.to(dt), {n: T(v, dt) for n, v in w.items()}


def subsample(a, step):
    return np.ascontiguousarray(a.reshape(-1)[::step])


def ncc_ref(y_true, y_pred):
    """reference NCC_vxm hard-codes .to('cuda') (losses.py:57): patch Tensor.to for the call."""
    orig = torch.Tensor.to

    def to(self, *a, **k):
        a = tuple("cpu" if (isinstance(x, str) and x == "cuda") else x for x in a)
        return orig(self, *a, **k)

    torch.Tensor.to = to
    try:
        ones = torch.ones
        torch.ones = lambda *a, **k: ones(*a, **{**k, "dtype": y_true.dtype})
        return ref_losses.NCC_vxm()(y_true, y_pred)
    finally:
        torch.Tensor.to = orig
        torch.ones = ones


# ----------------------------------------------------------------------------- end to end
def e2e(shape, tag, scale, stride, with_train):
    m, p = ref_model(shape, scale)
    mov, fix = synth.make_pair(shape, 24)
    mov, fix = T(mov), T(fix)
    for t in p.values():
        t.requires_grad_(True)
    y_ref, f_ref = m(mov, fix)
    taps = {}
    y_o, f_o = orc.modet_forward(p, mov, fix, HEADS, 6, scale, taps=taps)
    report(f"e2e[{tag}] flow fp64", f_o, f_ref)
    report(f"e2e[{tag}] y_moved fp64", y_o, y_ref)
    # fp32 noise floor of the reference itself
    m32, _ = ref_model(shape, scale, torch.float32)
    with torch.no_grad():
        y32, f32 = m32(mov.float(), fix.float())
    report(f"e2e[{tag}] REFERENCE fp32 vs fp64 flow (noise floor)", f32, f_ref)
    report(f"e2e[{tag}] REFERENCE fp32 vs fp64 y_moved (noise floor)", y32, y_ref)
    out = {
        "shape": np.array(shape), "scale": np.array(0.0 if scale is None else scale),
        "stride": np.array(stride),
        "flow": subsample(f_ref.detach().numpy().astype(np.float32), stride),
        "y_moved": subsample(y_ref.detach().numpy().astype(np.float32), stride),
        "flow_absmax": np.array(float(f_ref.abs().max())),
        "flow_sum": np.array(float(f_ref.sum())),
        "y_sum": np.array(float(y_ref.sum())),
    }
    # per-level sub-flow taps at stride (coarse levels are tiny: keep whole)
    for lvl in (5, 4, 3, 2, 1):
        out[f"w{lvl}"] = subsample(taps[f"w{lvl}"].detach().numpy().astype(np.float32),
                                   stride if lvl <= 2 else 1)
    if with_train:
        sim = ncc_ref(fix, y_ref)
        reg = ref_losses.Grad3d(penalty="l2")(f_ref, fix)
        loss = sim + reg
        names = list(p.keys())
        params = dict(m.named_parameters())
        g_ref = torch.autograd.grad(loss, [params[n] for n in names])
        l_o, s_o, r_o, _, _ = orc.train_loss(p, mov, fix, HEADS, 6, scale)
        g_o = torch.autograd.grad(l_o, [p[n] for n in names])
        report(f"train[{tag}] loss", l_o, loss)
        report(f"train[{tag}] ncc", s_o, sim)
        report(f"train[{tag}] grad3d", r_o, reg)
        out["loss"] = np.array([float(loss), float(sim), float(reg)])
        for n, gr, go in zip(names, g_ref, g_o):
            report(f"train[{tag}] grad {n}", go, gr)
            g = gr.detach().numpy()
            key = "grad." + n
            out[key] = g.astype(np.float64) if g.size <= 4096 else subsample(g, 61).astype(np.float64)
            out["gnorm." + n] = np.array(float(np.sqrt((g * g).sum())))
        # two Adam-amsgrad steps with the reference optimiser (train.py:101,:117)
        m2, _ = ref_model(shape, scale)
        opt = torch.optim.Adam(m2.parameters(), lr=1e-4, weight_decay=0, amsgrad=True)
        before = {n: q.detach().clone() for n, q in m2.named_parameters()}
        for it in range(2):
            for gpar in opt.param_groups:
                gpar["lr"] = round(1e-4 * np.power(1 - 0 / 30, 0.9), 8)
            y2, f2 = m2(mov, fix)
            l2 = ncc_ref(fix, y2) + ref_losses.Grad3d(penalty="l2")(f2, fix)
            opt.zero_grad()
            l2.backward()
            opt.step()
            out[f"adam_loss{it}"] = np.array(float(l2))
        for n, q in m2.named_parameters():
            d = (q.detach() - before[n]).numpy()
            out["delta." + n] = d if d.size <= 4096 else subsample(d, 61)
    np.savez_compressed(os.path.join(HERE, f"e2e_{tag}.npz"), **out)


# ----------------------------------------------------------------------------- per op
def op_attention():
    g = np.random.default_rng(101)
    out = {}
    for heads, shape, scale in ((1, (5, 6, 7), 1.0), (2, (4, 5, 3), None), (8, (3, 4, 5), 1.0), (4, (2, 3, 2), 1.0)):
        dim = 6 * heads
        mt = ref_models.ModeTransformer(dim, heads, qk_scale=scale).double()
        rpb = T(g.normal(0, 0.5, (heads, 3, 3, 3)))
        with torch.no_grad():
            mt.rpb.copy_(rpb)
        q = T(g.normal(0, 1, (2,) + shape + (dim,))).requires_grad_(True)
        k = T(g.normal(0, 1, (2,) + shape + (dim,))).requires_grad_(True)
        y = mt(q, k)
        gy = T(g.normal(0, 1, tuple(y.shape)))
        dq, dk, drpb = torch.autograd.grad((y * gy).sum(), [q, k, mt.rpb])
        sc = scale if scale else 6 ** -0.5
        rp = rpb.clone().requires_grad_(True)
        yo = orc.mode_transformer(q, k, rp, heads, sc)
        dqo, dko, dro = torch.autograd.grad((yo * gy).sum(), [q, k, rp])
        tag = f"h{heads}"
        report(f"attention[{tag}] out", yo, y)
        report(f"attention[{tag}] dq", dqo, dq)
        report(f"attention[{tag}] dk", dko, dk)
        report(f"attention[{tag}] drpb", dro, drpb)
        # logits as the CUDA op defines them (modet_kernel.cu:44-83) for the compat entry
        logits = orc.neighbourhood_logits(q, k, rpb, heads, sc)
        for nme, v in (("q", q), ("k", k), ("rpb", rpb), ("out", y), ("gy", gy), ("dq", dq), ("dk", dk),
                       ("drpb", drpb), ("logits", logits)):
            out[f"{tag}.{nme}"] = v.detach().numpy()
        out[f"{tag}.scale"] = np.array(sc)
    np.savez_compressed(os.path.join(HERE, "op_attention.npz"), **out)


def op_warp():
    g = np.random.default_rng(102)
    out = {}
    for tag, shape, C, amp in (("a", (6, 7, 5), 3, 2.5), ("b", (4, 4, 8), 8, 1.0), ("c", (3, 5, 4), 1, 6.0)):
        src = T(g.normal(0, 1, (2, C) + shape)).requires_grad_(True)
        flow = T(g.normal(0, amp, (2, 3) + shape)).requires_grad_(True)
        st = ref_models.SpatialTransformer(shape).double()
        y = st(src, flow)
        gy = T(g.normal(0, 1, tuple(y.shape)))
        ds, df = torch.autograd.grad((y * gy).sum(), [src, flow])
        yo = orc.warp(src, flow)
        dso, dfo = torch.autograd.grad((yo * gy).sum(), [src, flow])
        report(f"warp[{tag}] out", yo, y)
        report(f"warp[{tag}] dsrc", dso, ds)
        report(f"warp[{tag}] dflow", dfo, df)
        stn = ref_models.SpatialTransformer(shape, "nearest").double()
        lab = T(g.integers(0, 55, (2, 1) + shape).astype(np.float64))
        # keep nearest samples away from .5 ties (round-half-even vs fp noise of the round trip)
        fl_n = flow.detach().clone()
        frac = fl_n - torch.floor(fl_n)
        fl_n = torch.where((frac - 0.5).abs() < 1e-3, fl_n + 0.01, fl_n)
        yn = stn(lab, fl_n)
        report(f"warp[{tag}] nearest", orc.warp(lab, fl_n, "nearest"), yn)
        for nme, v in (("src", src), ("flow", flow), ("out", y), ("gy", gy), ("dsrc", ds), ("dflow", df),
                       ("lab", lab), ("flow_n", fl_n), ("out_n", yn)):
            out[f"{tag}.{nme}"] = v.detach().numpy()
    np.savez_compressed(os.path.join(HERE, "op_warp.npz"), **out)


def op_misc():
    g = np.random.default_rng(103)
    torch.manual_seed(103)      # the reference modules below draw their default init from torch's global generator
    out = {}
    # projection
    for tag, cin, dim, shape in (("p1", 8, 6, (3, 4, 5)), ("p5", 128, 48, (2, 3, 2)), ("p3", 32, 12, (3, 3, 4))):
        pl = ref_models.ProjectionLayer(cin, dim).double()
        p = {"x.proj.weight": T(g.normal(0, 0.3, (dim, cin))), "x.proj.bias": T(g.normal(0, 0.1, dim)),
             "x.norm.weight": T(1 + g.normal(0, 0.1, dim)), "x.norm.bias": T(g.normal(0, 0.1, dim))}
        with torch.no_grad():
            pl.proj.weight.copy_(p["x.proj.weight"]); pl.proj.bias.copy_(p["x.proj.bias"])
            pl.norm.weight.copy_(p["x.norm.weight"]); pl.norm.bias.copy_(p["x.norm.bias"])
        x = T(g.normal(0, 1, (2, cin) + shape)).requires_grad_(True)
        y = pl(x)
        gy = T(g.normal(0, 1, tuple(y.shape)))
        prm = [pl.proj.weight, pl.proj.bias, pl.norm.weight, pl.norm.bias]
        grads = torch.autograd.grad((y * gy).sum(), [x] + prm)
        for t in p.values():
            t.requires_grad_(True)
        yo = orc.projection(p, "x", x)
        go = torch.autograd.grad((yo * gy).sum(), [x] + list(p.values()))
        report(f"projection[{tag}] out", yo, y)
        for nme, a, b in zip(("dx", "dW", "db", "dgamma", "dbeta"), go, grads):
            report(f"projection[{tag}] {nme}", a, b)
        for nme, v in (("x", x), ("W", p["x.proj.weight"]), ("b", p["x.proj.bias"]), ("gamma", p["x.norm.weight"]),
                       ("beta", p["x.norm.bias"]), ("out", y), ("gy", gy), ("dx", grads[0]), ("dW", grads[1]),
                       ("db", grads[2]), ("dgamma", grads[3]), ("dbeta", grads[4])):
            out[f"{tag}.{nme}"] = v.detach().numpy()
    # conv blocks + pool
    for tag, cin, cout, shape, ins in (("c0", 1, 4, (5, 6, 7), False), ("c1", 4, 8, (4, 6, 5), True),
                                       ("c2", 16, 32, (3, 4, 4), True), ("c3", 24, 4, (4, 3, 5), True)):
        blk = (ref_models.ConvInsBlock if ins else ref_models.ConvBlock)(cin, cout).double()
        x = T(g.normal(0, 1, (2, cin) + shape)).requires_grad_(True)
        y = blk(x)
        gy = T(g.normal(0, 1, tuple(y.shape)))
        gx, gw, gb = torch.autograd.grad((y * gy).sum(), [x, blk.main.weight, blk.main.bias])
        p = {"b.main.weight": blk.main.weight.detach().clone().requires_grad_(True),
             "b.main.bias": blk.main.bias.detach().clone().requires_grad_(True)}
        yo = (orc.conv_ins_block if ins else orc.conv_block)(p, "b", x)
        gxo, gwo, gbo = torch.autograd.grad((yo * gy).sum(), [x, p["b.main.weight"], p["b.main.bias"]])
        report(f"conv[{tag}] out", yo, y); report(f"conv[{tag}] dx", gxo, gx)
        report(f"conv[{tag}] dw", gwo, gw); report(f"conv[{tag}] db", gbo, gb)
        yraw = torch.nn.functional.conv3d(x, blk.main.weight, blk.main.bias, padding=1)
        for nme, v in (("x", x), ("w", blk.main.weight), ("b", blk.main.bias), ("out", y), ("gy", gy), ("dx", gx),
                       ("dw", gw), ("db", gb), ("raw", yraw)):
            out[f"{tag}.{nme}"] = v.detach().numpy()
        out[f"{tag}.ins"] = np.array(ins)
    x = T(g.normal(0, 1, (2, 5, 4, 6, 8))).requires_grad_(True)
    y = torch.nn.AvgPool3d(2)(x)
    gy = T(g.normal(0, 1, tuple(y.shape)))
    out["pool.x"], out["pool.out"], out["pool.gy"] = x.detach().numpy(), y.detach().numpy(), gy.numpy()
    out["pool.dx"] = torch.autograd.grad((y * gy).sum(), x)[0].numpy()
    # upsample (as ModeT uses it: nn.Upsample(2,'trilinear',align_corners=True) on 2*flow)
    up = torch.nn.Upsample(scale_factor=2, mode="trilinear", align_corners=True)
    x = T(g.normal(0, 1, (2, 3, 3, 4, 5))).requires_grad_(True)
    y = up(2 * x)
    gy = T(g.normal(0, 1, tuple(y.shape)))
    report("upsample out", orc.upsample2(2 * x), y)
    out["up.x"], out["up.out"], out["up.gy"] = x.detach().numpy(), y.detach().numpy(), gy.numpy()
    out["up.dx"] = torch.autograd.grad((y * gy).sum(), x)[0].numpy()
    # CWM
    for tag, heads, shape in (("w3", 2, (3, 4, 3)), ("w5", 8, (2, 3, 2))):
        c = 3 * heads
        mod = ref_models.CWM(c, 2 * c).double()
        x = T(g.normal(0, 0.6, (2, c) + shape)).requires_grad_(True)
        y = mod(x)
        gy = T(g.normal(0, 1, tuple(y.shape)))
        prm = list(mod.parameters())
        grads = torch.autograd.grad((y * gy).sum(), [x] + prm)
        p = {"m." + n: q.detach().clone().requires_grad_(True) for n, q in mod.named_parameters()}
        yo = orc.cwm(p, "m", x, heads)
        go = torch.autograd.grad((yo * gy).sum(), [x] + list(p.values()))
        report(f"cwm[{tag}] out", yo, y)
        for (n, _), a, b in zip([("x", None)] + list(mod.named_parameters()), go, grads):
            report(f"cwm[{tag}] d{n}", a, b)
        out[f"{tag}.x"], out[f"{tag}.out"], out[f"{tag}.gy"] = x.detach().numpy(), y.detach().numpy(), gy.numpy()
        out[f"{tag}.dx"] = grads[0].numpy()
        for (n, q), gq in zip(mod.named_parameters(), grads[1:]):
            out[f"{tag}.p.{n}"] = q.detach().numpy()
            out[f"{tag}.g.{n}"] = gq.numpy()
    # losses
    shape = (12, 14, 11)
    a = T(synth.make_volume(shape, 5).astype(np.float64))[None, None].requires_grad_(True)
    b = T(synth.make_volume(shape, 6).astype(np.float64))[None, None].requires_grad_(True)
    l = ncc_ref(a, b)
    ga, gb = torch.autograd.grad(l, [a, b])
    lo = orc.ncc_loss(a, b)
    gao, gbo = torch.autograd.grad(lo, [a, b])
    report("ncc value", lo, l); report("ncc d y_true", gao, ga); report("ncc d y_pred", gbo, gb)
    out["ncc.a"], out["ncc.b"], out["ncc.val"] = a.detach().numpy(), b.detach().numpy(), np.array(float(l))
    out["ncc.da"], out["ncc.db"] = ga.numpy(), gb.numpy()
    fl = T(g.normal(0, 1, (2, 3, 5, 6, 4))).requires_grad_(True)
    l = ref_losses.Grad3d(penalty="l2")(fl, None)
    gf = torch.autograd.grad(l, fl)[0]
    report("grad3d value", orc.grad3d_loss(fl), l)
    out["g3d.flow"], out["g3d.val"], out["g3d.dflow"] = fl.detach().numpy(), np.array(float(l)), gf.numpy()
    np.savez_compressed(os.path.join(HERE, "op_misc.npz"), **out)


def op_dice():
    """nearest label warp + Dice (utils.py:74-106; register_model needs .cuda(), so the
    label warp uses models.SpatialTransformer(size,'nearest'): identical math, models.py:25-67)."""
    shape = (32, 48, 32)
    lab_m = synth.make_labels(shape, 24).astype(np.float64)
    lab_f = synth.make_labels(shape, 25).astype(np.float64)
    flow = synth.make_flow(shape, 3, 2.0).astype(np.float64)
    frac = flow - np.floor(flow)
    flow = np.where(np.abs(frac - 0.5) < 1e-3, flow + 0.01, flow)
    st = ref_models.SpatialTransformer(shape, "nearest").double()
    warped = st(T(lab_m)[None, None], T(flow))
    VOI = list(range(1, 55))
    pred, true = warped.long().numpy()[0, 0], lab_f.astype(np.int64)
    dscs = []
    for i in VOI:  # arithmetic of utils.py:95-105
        pi, ti = pred == i, true == i
        dscs.append(2.0 * np.sum(pi * ti) / (np.sum(pi) + np.sum(ti) + 1e-5))
    d = float(np.mean(dscs))
    do = orc.dice_voi(orc.warp(T(lab_m)[None, None], T(flow), "nearest").long(), T(lab_f)[None, None].long())
    REPORT.append(f"{'dice nearest-warp':58s} |oracle-ref| = {abs(do - d):.3e}   ref = {d:.6f}")
    raw = float(np.mean([2.0 * np.sum((lab_m == i) * (lab_f == i)) / (np.sum(lab_m == i) + np.sum(lab_f == i) + 1e-5)
                         for i in VOI]))
    np.savez_compressed(os.path.join(HERE, "op_dice.npz"), shape=np.array(shape), flow=flow.astype(np.float32),
                        warped=pred.astype(np.int16), dice=np.array(d), dice_raw=np.array(raw))


if __name__ == "__main__":
    op_attention()
    op_warp()
    op_misc()
    op_dice()
    e2e((32, 48, 32), "32x48x32", 1.0, 1, True)      # one dim < 48: L5 smaller than the window
    e2e((48, 64, 48), "48x64x48", None, 5, False)    # scale=None -> head_dim**-0.5
    with open(os.path.join(HERE, "REPORT.txt"), "w") as f:
        f.write("oracle/modet_torch.py vs /root/reference/ModeT (fp64, CPU, torch %s)\n" % torch.__version__)
        f.write("\n".join(REPORT) + "\n")
    print("\n".join(REPORT))
