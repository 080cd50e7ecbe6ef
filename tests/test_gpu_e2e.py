"""-m gpu: the whole hot path (ModeT.forward, the train step, the eval tail) on the MI355X against
the reference's golden vectors (fp64) and the CPU oracle.  Tolerances follow SURVEY.md §8(c):
the reference's OWN fp32 run deviates from its fp64 run by 4e-5..9e-5 voxels on these fixtures
(tests/golden/REPORT.txt), we accept max|flow err| <= 2e-3 voxels, y_moved <= 5e-5... (measured values
are written to gpurun_out/parity_report.json)."""
import os

import numpy as np
import pytest
import torch

from tests.util import assert_close, gold, np64
from tests.util import note as _note

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _model(shape, scale, cls="ModeT"):
    from smilecode_amd import models, synth
    m = getattr(models, cls)(shape, head_dim=6, num_heads=[8, 4, 2, 1, 1], scale=scale).cuda()
    models.load_numpy_weights(m, synth.make_weights(24))
    return m


def _pair(shape, batch=1):
    from smilecode_amd import synth
    mov, fix = synth.make_pair(shape, 24, batch)
    return torch.from_numpy(mov).cuda(), torch.from_numpy(fix).cuda()


@pytest.mark.parametrize("tag,scale", [("32x48x32", 1.0), ("48x64x48", None)])
def test_forward_golden(tag, scale):
    g = gold(f"e2e_{tag}.npz")
    shape = tuple(int(s) for s in g["shape"])
    stride = int(g["stride"])
    model = _model(shape, scale)
    mov, fix = _pair(shape)
    with torch.no_grad():
        y, flow = model(mov, fix)
    assert tuple(y.shape) == (1, 1) + shape and tuple(flow.shape) == (1, 3) + shape
    ef = assert_close(np64(flow).reshape(-1)[::stride], g["flow"], atol=2e-3, rtol=0, what="flow vs reference fp64")
    ey = assert_close(np64(y).reshape(-1)[::stride], g["y_moved"], atol=5e-5, rtol=0, what="y_moved vs reference fp64")
    _note(f"fwd[{tag}].flow_maxerr_voxels", ef)
    _note(f"fwd[{tag}].y_moved_maxerr", ey)
    _note(f"fwd[{tag}].flow_absmax", float(g["flow_absmax"]))


@pytest.mark.parametrize("shape,B", [((48, 64, 48), 1), ((48, 48, 64), 2)])
def test_modet_cu_through_the_operator_boundary(shape, B):
    """VERDICT r2 missing-2: ``ModeT_cu(fused_attention=False)`` runs every level's attention the way the reference's
    ModeT-cu does -- layout prep -> ``modetqkrpb_cu`` -> softmax -> ``@ v`` (ModeT-cu/models.py:300-316) -- so the
    operator kernels (qk_*_plane_kernel) see the model's real q / k / d_attn at every pyramid level (heads 8/4/2/1/1).
    Flow, loss and every parameter gradient must equal the fused-kernel model's and the fp64 oracle's."""
    from oracle import modet_torch as orc
    from smilecode_amd import losses, models, synth
    w = synth.make_weights(24)
    res = {}
    mov_np, fix_np = synth.make_pair(shape, 24, B)
    mov, fix = torch.from_numpy(mov_np).cuda(), torch.from_numpy(fix_np).cuda()
    for fused in (True, False):
        m = models.ModeT_cu(shape, head_dim=6, num_heads=[8, 4, 2, 1, 1], scale=1, fused_attention=fused).cuda()
        models.load_numpy_weights(m, w)
        y, flow = m(mov, fix)
        loss = losses.NCC_vxm()(fix, y) + losses.Grad3d(penalty="l2")(flow, fix)
        loss.backward()
        res[fused] = (flow.detach(), float(loss), {n: p.grad.clone() for n, p in m.named_parameters()})
    # two fp32 evaluations of the same mathematics: fp32 noise of the pipeline (1e-4 .. 8e-4 voxels on |flow| ~ 9)
    ef = float((res[True][0] - res[False][0]).abs().max())
    assert ef < 1e-3 and abs(res[True][1] - res[False][1]) < 2e-6, (ef, res[True][1], res[False][1])
    p64 = {n: torch.from_numpy(v).double().requires_grad_(True) for n, v in w.items()}
    l64, _, _, _, f64 = orc.train_loss(p64, torch.from_numpy(mov_np).double(), torch.from_numpy(fix_np).double(),
                                       (8, 4, 2, 1, 1), 6, 1.0)
    g64 = dict(zip(p64, torch.autograd.grad(l64, list(p64.values()))))
    eo = float((res[False][0].double().cpu() - f64.detach()).abs().max())
    assert eo <= 2e-3 and abs(res[False][1] - float(l64)) < 2e-5, (eo, res[False][1], float(l64))
    worst = 0.0
    for n, ref in g64.items():
        gmax = float(ref.abs().max())
        if gmax < 1e-8:
            continue
        worst = max(worst, float((res[False][2][n].double().cpu() - ref).abs().max()) / gmax)
    assert worst <= 5e-3, worst                      # (measured 2.6e-4 / 1.8e-3; round 5's bound was 2e-2: VERDICT r5 item 2)
    _note(f"modet_cu_operator[{'x'.join(map(str, shape))},B={B}].flow_maxerr_vs_fp64", eo)
    _note(f"modet_cu_operator[{'x'.join(map(str, shape))},B={B}].flow_maxdiff_vs_fused", ef)
    _note(f"modet_cu_operator[{'x'.join(map(str, shape))},B={B}].grad_worst_rel_to_max", worst)


def test_modet_cu_operator_path_rejects_levels_below_the_window():
    """CHECK_3DFEATMAP (ModeT-cu/modet/include/utils.h:10): with 32x48x32 the coarsest level is 2x3x2 < 3x3x3 and the
    reference's extension raises; so does ours (the fused kernel handles it, as the pure-PyTorch ModeT does)."""
    from smilecode_amd import models
    shape = (32, 48, 32)
    m = models.ModeT_cu(shape, fused_attention=False).cuda()
    mov, fix = _pair(shape)
    with pytest.raises(RuntimeError, match="greater than or equal to kernel size"):
        m(mov, fix)


def test_inshape_not_divisible_by_16_fails_in_forward_like_the_reference():
    """the reference's constructor accepts any inshape (models.py:338-375) and its forward raises a RuntimeError when the
    x2-upsampled flow no longer matches the next level; same here: construction succeeds, forward raises"""
    from smilecode_amd import models
    m = models.ModeT((40, 48, 40)).cuda()
    mov, fix = _pair((40, 48, 40))
    with pytest.raises(RuntimeError, match="multiple of 16"):
        m(mov, fix)


def test_modet_cu_same_network_and_state_dict_roundtrip():
    from smilecode_amd import models
    shape = (32, 48, 32)
    a = _model(shape, 1.0, "ModeT")
    b = models.ModeT_cu(shape).cuda()
    sd = a.state_dict()
    assert "mdt1.grid" in sd and not any(k.startswith("transformer") for k in sd)
    # a reference checkpoint carries transformer.N.grid and mdtN.grid: must load strictly into either class
    sd["transformer.0.grid"] = torch.zeros(1, 3, *shape)
    b.load_state_dict(sd, strict=True)
    assert "mdt1.v" in b.state_dict()
    mov, fix = _pair(shape)
    with torch.no_grad():
        ya, fa = a(mov, fix)
        yb, fb = b(mov, fix)
    assert torch.equal(fa, fb) and torch.equal(ya, yb)


def test_train_step_golden():
    """loss, every parameter gradient and two Adam-amsgrad steps vs the reference's fp64 autograd."""
    from smilecode_amd.engine import Trainer
    g = gold("e2e_32x48x32.npz")
    shape = (32, 48, 32)
    model = _model(shape, 1.0)
    mov, fix = _pair(shape)
    tr = Trainer(model)
    before = tr.fp.flat.clone()
    tr.fp.zero_grad()
    loss, sim, reg = tr.loss(mov, fix)
    loss.backward()
    assert_close(np.array([float(loss), float(sim), float(reg)]), g["loss"], atol=2e-4, rtol=1e-4, what="loss/ncc/grad3d")
    worst = 0.0
    for name, p in model.named_parameters():
        ref = g["grad." + name]
        got = np64(p.grad).reshape(-1)
        got = got if ref.size == got.size else got[::61]
        gmax = float(np.abs(ref).max())
        if name.endswith("main.bias") and gmax < 1e-9:      # conv bias under InstanceNorm: analytically zero
            assert float(np.abs(got).max()) < 1e-5, name
            continue
        err = float(np.abs(got - ref.reshape(-1)).max())
        worst = max(worst, err / gmax)
        assert err <= 5e-3 * gmax + 1e-7, f"grad {name}: max err {err:.3e} vs max|g| {gmax:.3e}"
    _note("train.grad_worst_rel_to_max", worst)
    # two optimizer steps (train.py:131-133)
    model2 = _model(shape, 1.0)
    tr2 = Trainer(model2)
    for it in range(2):
        l, _, _ = tr2.train_step(mov, fix, epoch=0)
        assert abs(float(l) - float(g[f"adam_loss{it}"])) < 5e-4
    wd = 0.0
    for name, p in model2.named_parameters():
        ref = g["delta." + name].reshape(-1)
        off, k = tr2.fp.offsets[[n for n, _ in model2.named_parameters()].index(name)]
        got = np64(p.detach().reshape(-1) - before[off:off + k])
        got = got if ref.size == got.size else got[::61]
        if name.endswith("main.bias") and "conv.2" not in name and "conv0.0" not in name:
            continue      # zero-gradient biases: Adam normalises pure noise to +-lr, sign is arbitrary
        # Adam's update is ~ +-lr per step wherever |g| >> eps; compare where the reference moved decisively
        assert np.abs(got).max() <= 2.1e-4
        wd = max(wd, float(np.abs(got - ref).max()))
    _note("train.adam_delta_maxerr", wd)


def test_forward_vs_oracle_64_batch2():
    """cfg-1 shape (64^3) with B=2 against the CPU oracle in fp32 (independent ATen-CPU arithmetic)."""
    from oracle import modet_torch as orc
    from smilecode_amd import synth
    shape = (64, 64, 64)
    model = _model(shape, 1.0)
    mov, fix = _pair(shape, 2)
    with torch.no_grad():
        y, flow = model(mov, fix)
    p = {n: torch.from_numpy(v).double() for n, v in synth.make_weights(24).items()}
    with torch.no_grad():
        yr, fr = orc.modet_forward(p, mov.double().cpu(), fix.double().cpu(), (8, 4, 2, 1, 1), 6, 1.0)
    ef = assert_close(np64(flow), fr.numpy(), atol=2e-3, rtol=0, what="flow vs oracle fp64")
    # y_moved inherits |grad(moving)| * flow error (image gradients reach ~0.5/voxel at the mask edge)
    ey = assert_close(np64(y), yr.numpy(), atol=5e-4, rtol=0, what="y_moved vs oracle fp64")
    _note("fwd[64^3,B=2].flow_maxerr_voxels", ef)
    _note("fwd[64^3,B=2].y_moved_maxerr", ey)


@pytest.mark.parametrize("gain", [4000.0, 60000.0])
def test_forward_and_grads_with_an_unnormalised_image(gain):
    """The reference runs on whatever intensities it is given (nn.Conv3d has no range, reference models.py:119-133): a user who
    forgot to min-max an MR volume (x 4 000) or feeds raw 16-bit intensities (x 60 000) must get the reference's answer, not
    NaN (round 5's f16 forward forms overflowed in the first ConvInsBlock).  Forward and the gradients of one
    loss against the fp64 oracle on the same scaled pair; the ConvBlock 1 -> 4 output reaches ~1e4 / ~2e5 here."""
    from oracle import modet_torch as orc
    from smilecode_amd import losses, synth
    shape = (32, 48, 32)
    w = synth.make_weights(24)
    mov_np, fix_np = (a * np.float32(gain) for a in synth.make_pair(shape, 24))
    p64 = {n: torch.from_numpy(v).double().requires_grad_(True) for n, v in w.items()}
    loss64, _, _, y64, f64 = orc.train_loss(p64, torch.from_numpy(mov_np).double(), torch.from_numpy(fix_np).double(), (8, 4, 2, 1, 1), 6, 1.0)
    g64 = torch.autograd.grad(loss64, list(p64.values()), allow_unused=True)
    model = _model(shape, 1.0)
    mov, fix = torch.from_numpy(mov_np).cuda(), torch.from_numpy(fix_np).cuda()
    y, flow = model(mov, fix)
    assert bool(torch.isfinite(flow).all()) and bool(torch.isfinite(y).all())
    _note(f"unnormalised[x{gain:g}].flow_maxerr", assert_close(np64(flow), f64.detach().numpy(), atol=2e-3, rtol=0, what="flow (voxels)"))
    assert_close(np64(y) / gain, y64.detach().numpy() / gain, atol=5e-4, rtol=0, what="y_moved / gain")
    loss = losses.NCC_vxm()(fix, y) + losses.Grad3d(penalty="l2")(flow, fix)
    assert abs(float(loss.detach()) - float(loss64.detach())) < 2e-4 * max(1.0, abs(float(loss64.detach())))
    loss.backward()
    worst = 0.0
    for (n, p_), g in zip(model.named_parameters(), g64):
        if g is None or float(g.abs().max()) < 1e-8 * max(1.0, float(max(x.abs().max() for x in g64 if x is not None))):
            continue                                           # (analytically zero: conv biases under InstanceNorm)
        assert p_.grad is not None and bool(torch.isfinite(p_.grad).all()), n
        e = float((p_.grad.double().cpu() - g).abs().max() / g.abs().max())
        worst = max(worst, e)
    _note(f"unnormalised[x{gain:g}].worst_grad_relerr", worst)
    assert worst < 5e-3, worst


@pytest.mark.parametrize("shape,batch", [((16, 32, 16), 1), ((32, 32, 48), 3), ((16, 48, 80), 2)])
def test_forward_and_grads_vs_oracle_edge_shapes(shape, batch):
    """smallest volume the reference's pure path accepts (16x32x16: level 5 is 1x2x1, smaller than the 3^3 window, every
    neighbour of most voxels is zero padding), an odd batch, and a strongly anisotropic volume: forward and the full
    gradient dict of one loss against the fp64 oracle."""
    from oracle import modet_torch as orc
    from smilecode_amd import losses, synth
    w = synth.make_weights(24)
    mov_np, fix_np = synth.make_pair(shape, 31, batch)
    p64 = {n: torch.from_numpy(v).double().requires_grad_(True) for n, v in w.items()}
    heads = (8, 4, 2, 1, 1)
    loss64, _, _, y64, f64 = orc.train_loss(p64, torch.from_numpy(mov_np).double(), torch.from_numpy(fix_np).double(), heads, 6, 1.0)
    g64 = torch.autograd.grad(loss64, list(p64.values()))
    y64, f64 = y64.detach(), f64.detach()
    model = _model(shape, 1.0)
    mov, fix = torch.from_numpy(mov_np).cuda(), torch.from_numpy(fix_np).cuda()
    y, flow = model(mov, fix)
    tag = "x".join(map(str, shape)) + f"_B{batch}"
    _note(f"edge_{tag}_flow_maxerr", assert_close(np64(flow), f64.numpy(), atol=2e-3, rtol=0, what="flow (voxels)"))
    assert_close(np64(y), y64.numpy(), atol=5e-4, rtol=0, what="y_moved")
    loss = losses.NCC_vxm()(fix, y) + losses.Grad3d(penalty="l2")(flow, fix)
    assert abs(float(loss.detach()) - float(loss64.detach())) < 2e-6
    loss.backward()
    worst = 0.0
    for (n, prm), gref in zip(model.named_parameters(), g64):
        g = prm.grad.double().cpu()
        scale = max(float(gref.abs().max()), 1e-7)
        worst = max(worst, float((g - gref).abs().max()) / scale if float(gref.abs().max()) > 1e-6 else 0.0)
    _note(f"edge_{tag}_grad_relerr", worst)
    assert worst < 5e-3, worst


def test_full_size_properties():
    """BASELINE size 160x192x160: size-independent properties of the forward."""
    shape = (160, 192, 160)
    model = _model(shape, 1.0)
    mov, fix = _pair(shape)
    with torch.no_grad():
        y1, f1 = model(mov, fix)
        y2, f2 = model(mov, fix)
    assert torch.isfinite(f1).all() and torch.isfinite(y1).all()
    assert torch.equal(f1, f2) and torch.equal(y1, y2), "forward must be run-to-run deterministic"
    # y_moved is exactly warp(moving, flow) through the public SpatialTransformer (models.py:410)
    from smilecode_amd.models import SpatialTransformer
    y3 = SpatialTransformer(shape)(mov, f1)
    assert torch.equal(y3, y1)
    # identical images + zeroed rpb/proj -> uniform attention -> zero flow -> identity warp
    from smilecode_amd import models
    m0 = models.ModeT(shape, scale=1).cuda()
    with torch.no_grad():
        for n, p in m0.named_parameters():
            if "proj.weight" in n or "rpb" in n:
                p.zero_()
        y0, f0 = m0(mov, mov)
    assert float(f0.abs().max()) < 1e-5
    assert float((y0 - mov).abs().max()) < 1e-5
    _note("fwd[160x192x160].flow_absmax", float(f1.abs().max()))


def test_full_size_train_step_runs_and_is_sane():
    from smilecode_amd.engine import Trainer
    shape = (160, 192, 160)
    model = _model(shape, 1.0)
    mov, fix = _pair(shape)
    tr = Trainer(model)
    l0, s0, r0 = tr.train_step(mov, fix)
    assert torch.isfinite(tr.fp.grad).all()
    assert float(tr.fp.grad.abs().max()) > 0          # flat buffer holds the packed gradients of the last step
    for _ in range(3):
        l1, s1, r1 = tr.train_step(mov, fix)
    assert torch.isfinite(l1)
    assert float(l1) < float(l0) + 1e-3, "loss should not blow up over 4 Adam steps on one pair"
    _note("train[160x192x160].loss0", float(l0))
    _note("train[160x192x160].loss3", float(l1))


def test_full_size_dice_parity_vs_oracle(oracle_job):
    """north-star parity statement: Dice on (synthetic 54-label) LPBA-shaped labels matches the reference path to
    +-0.001 at 160x192x160.  Reference path = CPU oracle forward (fp32, same op sequence as ModeT/models.py) ->
    nearest label warp -> dice_val_VOI arithmetic (utils.py:86-106); ours = HIP forward -> fused GPU eval tail.
    The oracle's full-size runs (fp32 and fp64 flow) come from tests/oracle_jobs.py, computed once beside the GPU tests."""
    from oracle import modet_torch as orc
    from smilecode_amd import synth
    from smilecode_amd.utils import warp_labels_and_dice
    shape = (160, 192, 160)
    model = _model(shape, 1.0)
    mov, fix = _pair(shape)
    with torch.no_grad():
        _, flow = model(mov, fix)
    lab_m = torch.from_numpy(synth.make_labels(shape, 24))[None, None]
    lab_f = torch.from_numpy(synth.make_labels(shape, 25))[None, None]
    warped, dice_gpu = warp_labels_and_dice(lab_m.cuda(), flow, lab_f.cuda())
    # (1) same flow through the oracle's eval tail: the label warp must agree voxel for voxel except exact .5 ties
    w_ref = orc.warp(lab_m.float(), flow.cpu(), "nearest")
    mism = float((w_ref.to(torch.int16) != warped.cpu()).float().mean())
    assert mism < 1e-5, f"label warp differs on {mism:.2e} of the voxels"
    d_same = orc.dice_voi(w_ref.long(), lab_f.long())
    assert abs(d_same - dice_gpu) < 1e-4
    # (2) the oracle's own forward (CPU fp32) -> its flow -> its Dice
    o = oracle_job("full160")
    f_ref, f64 = o["flow32"], o["flow64"]
    d_ref = orc.dice_voi(orc.warp(lab_m.float(), f_ref, "nearest").long(), lab_f.long())
    # fp64 oracle: separates our error from the fp32 CPU path's own (both are fp32 noise on |flow| up to ~17)
    e_hip = float((flow.double().cpu() - f64).abs().max())
    e_cpu32 = float((f_ref.double() - f64).abs().max())
    _note("fwd[160x192x160].flow_maxerr_voxels_hip_vs_fp64", e_hip)
    _note("fwd[160x192x160].flow_maxerr_voxels_cpu_fp32_vs_fp64", e_cpu32)
    _note("fwd[160x192x160].flow_rmserr_voxels_hip_vs_fp64", float((flow.double().cpu() - f64).pow(2).mean().sqrt()))
    # the stated end-to-end tolerance is max |flow error| <= 2e-3 voxels.  Asserted here since round 6
    # (VERDICT r5 item 2): 8e-4 -- measured 4.3e-4, >= 45 % headroom (round 5: 1.20e-3; the un-scaled f16 pieces of the 4 -> 8
    # layer were the noise) -- and not noisier than twice the reference's own fp32 arithmetic on the same pair (4.7e-4)
    assert e_hip <= 8e-4, f"HIP flow deviates {e_hip:.2e} voxels from the fp64 oracle (fp32 CPU path: {e_cpu32:.2e})"
    assert e_hip <= 2.0 * e_cpu32, f"HIP flow error {e_hip:.2e} > 2 x the ATen-CPU fp32 path's {e_cpu32:.2e}"
    _note("dice[160x192x160].hip", dice_gpu)
    _note("dice[160x192x160].oracle", d_ref)
    _note("dice[160x192x160].flow_maxerr_voxels_fp32_vs_fp32", float((flow.cpu() - f_ref).abs().max()))
    assert abs(d_ref - dice_gpu) <= 1e-3, f"Dice {dice_gpu:.5f} vs reference path {d_ref:.5f}"


def test_full_size_gradient_parity_vs_oracle(oracle_job):
    """BASELINE size 160x192x160: loss and EVERY parameter gradient of one train step against the fp64 CPU oracle's
    autograd (ModeT/train.py:122-131 on ModeT/models.py:377-412), same tolerance as the small shapes: worst error per
    tensor <= 5e-3 of that tensor's max |g| (biases in front of an InstanceNorm have an analytically zero gradient and
    are compared absolutely)."""
    from smilecode_amd import losses, synth
    shape = (160, 192, 160)
    o = oracle_job("full160")
    g64, loss64, sim64, reg64 = o["grad"], o["loss"], o["sim"], o["reg"]
    mov_np, fix_np = synth.make_pair(shape, 24)
    model = _model(shape, 1.0)
    mov, fix = torch.from_numpy(mov_np).cuda(), torch.from_numpy(fix_np).cuda()
    y, flow = model(mov, fix)
    sim = losses.NCC_vxm()(fix, y)
    reg = losses.Grad3d(penalty="l2")(flow, fix)
    loss = sim + reg
    assert abs(float(sim.detach()) - float(sim64)) < 2e-5 and abs(float(reg.detach()) - float(reg64)) < 2e-6
    loss.backward()
    worst, worst_name = 0.0, ""
    for n, prm in model.named_parameters():
        ref = g64[n]
        gmax = float(ref.abs().max())
        err = float((prm.grad.double().cpu() - ref).abs().max())
        if gmax < 1e-8:                                     # conv bias under InstanceNorm: analytically zero
            assert err < 1e-5, (n, err)
            continue
        if err / gmax > worst:
            worst, worst_name = err / gmax, n
    _note("train[160x192x160].grad_worst_rel_to_max", worst)
    _note("train[160x192x160].loss_abs_err", abs(float(loss.detach()) - float(loss64)))
    assert worst <= 5e-3, f"worst gradient error {worst:.3e} of max|g| in {worst_name}"        # (measured 6.4e-4)


@pytest.mark.parametrize("overlap,graph", [("0", "0"), ("1", "0"), ("1", "1"), ("0", "1")])
def test_data_parallel_two_ranks_equal_batch_two(tmp_path, overlap, graph):
    """N ranks x 1 pair with one averaged all-reduce == one process with batch N (N = 2 ranks sharing
    this GPU over gloo; on the 8-GPU node the same code runs over RCCL).  overlap=1: the backward in three autograd stages,
    bucket k's all-reduce launched after stage k (cfg 5) instead of one all-reduce after backward; graph=1: the step
    replayed from hipGraphs (overlap: three stage graphs with the all-reduces between the replays)."""
    import subprocess
    import sys
    from smilecode_amd import models, synth
    from smilecode_amd.engine import Trainer
    shape = (32, 48, 32)
    env = dict(os.environ, MODET_DIST_BACKEND="gloo", HSA_ENABLE_IPC_MODE_LEGACY="0", MODET_OVERLAP=overlap, MODET_GRAPH=graph)
    port = 29700 + os.getpid() % 200 + int(overlap) + 2 * int(graph)
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr",
           "127.0.0.1", "--master-port", str(port), os.path.join(ROOT, "tests", "dp_worker.py"), str(tmp_path),
           ",".join(map(str, shape))]
    r = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=280)
    assert r.returncode == 0, r.stderr[-2000:]
    ranks = [np.load(tmp_path / f"rank{i}.npz") for i in range(2)]
    assert np.array_equal(ranks[0]["flat"], ranks[1]["flat"]), "ranks must hold identical parameters after the step"
    assert np.array_equal(ranks[0]["grad"], ranks[1]["grad"])
    # single process, batch 2 (the loss is a mean over the batch, so its gradient is the rank average)
    model = _model(shape, 1.0)
    tr = Trainer(model)
    mov, fix = _pair(shape, 2)
    tr.train_step(mov, fix, epoch=0)
    g1, f1 = tr.fp.grad.cpu().numpy(), tr.fp.flat.cpu().numpy()
    gerr = np.abs(ranks[0]["grad"] - g1).max() / np.abs(g1).max()
    _note(f"dp2_vs_batch2_grad_relerr[overlap={overlap},graph={graph}]", gerr)
    if not gerr < 2e-5:
        # name the side and the tensors: the batch-2 reference a second time (fresh trainer), then every parameter tensor
        tr2 = Trainer(_model(shape, 1.0))
        tr2.train_step(mov, fix, epoch=0)
        g2 = tr2.fp.grad.cpu().numpy()
        gm = np.abs(g1).max()
        lines = ["overlap=%s graph=%s: ranks vs reference %.3e, reference run 1 vs run 2 %.3e, ranks vs reference run 2 %.3e (of max|g|)"
                 % (overlap, graph, gerr, np.abs(g1 - g2).max() / gm, np.abs(ranks[0]["grad"] - g2).max() / gm)]
        for (n, _), (off, k) in zip(model.named_parameters(), tr.fp.offsets):
            sl = slice(off, off + k)
            lines.append("  %-36s %8d  ranks-ref %.3e  ref1-ref2 %.3e  (own max %.3e)" % (
                n, k, np.abs(ranks[0]["grad"][sl] - g1[sl]).max() / gm, np.abs(g1[sl] - g2[sl]).max() / gm, np.abs(g1[sl]).max() / gm))
        os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
        with open(os.path.join(ROOT, "gpurun_out", f"dp_diag_{overlap}{graph}.txt"), "w") as f:
            f.write("\n".join(lines) + "\n")
        print("\n".join(lines))
    assert gerr < 2e-5, gerr
    # the first Adam step moves every parameter by lr * g/(|g| + eps) ~ +-1e-4: compare where the gradient is not pure
    # rounding noise (biases in front of an InstanceNorm have analytically zero gradient, their sign is arbitrary)
    sig = np.abs(g1) > 1e-4 * np.abs(g1).max()
    assert sig.mean() > 0.5
    assert np.abs(ranks[0]["flat"] - f1)[sig].max() < 2e-5


@pytest.mark.parametrize("graph", ["0", "1"])
def test_staged_step_through_the_nccl_backend(tmp_path, graph):
    """the overlapped step (three stages, bucket k's all-reduce issued after stage k) on the `nccl`
    backend = RCCL, whose collectives are stream-ordered work on RCCL's own stream -- ordering semantics the two-rank gloo
    test cannot show.  One rank (this box has one GPU), the three collectives really issued; three consecutive steps must
    be the plain eager trainer's steps: losses, gradients, parameters."""
    import subprocess
    import sys
    from smilecode_amd.engine import Trainer
    shape = (32, 48, 32)
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0", MASTER_ADDR="127.0.0.1",
               MASTER_PORT=str(29900 + os.getpid() % 90 + int(graph)))
    cmd = [sys.executable, os.path.join(ROOT, "tests", "nccl_worker.py"), str(tmp_path), ",".join(map(str, shape)), graph, "3"]
    r = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=280)
    assert r.returncode == 0, r.stderr[-2000:]
    z = np.load(tmp_path / "nccl.npz")
    assert int(z["n_collectives"]) == 9, "three bucket all-reduces per step must have been issued"
    tr = Trainer(_model(shape, 1.0))
    mov, fix = _pair(shape)
    for step in range(3):
        out = tr.train_step(mov, fix, epoch=0)
        g = tr.fp.grad.cpu().numpy()
        gerr = np.abs(z["grads"][step] - g).max() / np.abs(g).max()
        _note(f"nccl_staged[graph={graph}].step{step}_grad_relerr", gerr)
        # step 0 runs on identical parameters; later steps on parameters that differ by Adam's unit-size response to the
        # float-atomic noise of step 0 (see test_hip_graph_training_trajectory_equals_eager)
        assert gerr < (2e-5 if step == 0 else 5e-2), (step, gerr)
        assert abs(float(out[0]) - float(z["losses"][step][0])) < (1e-6 if step == 0 else 4e-3)
    assert np.abs(z["flat"] - tr.fp.flat.cpu().numpy()).max() < 1e-3


def test_attention_backward_is_bit_stable_beside_another_process(tmp_path):
    """Round 5: with the SLP vectoriser's packed-fp32 pairs behind its 27 exps, na_bwd_kernel returned wrong d_q /
    d_rpb in lanes 48..63 of a wave about once per 100 launches WHEN ANOTHER PROCESS time-shares the GPU and a few
    high-priority streams exist in both -- the two-ranks-on-one-GPU test's set-up, and the reason it was red in round 4.  The
    kernel has no atomics: 6000 calls on fixed inputs beside such a process must be bit-identical (the -O3 build had ~60
    that were not; the library is built with -fno-slp-vectorize since)."""
    import subprocess
    import sys
    import time
    from smilecode_amd import _lib, ops
    D, H, W = 32, 48, 32
    g = torch.Generator().manual_seed(3)
    q, k = (torch.randn(1, D, H, W, 6, generator=g).cuda() for _ in range(2))
    rpb = (torch.randn(1, 3, 3, 3, generator=g) * 0.3).cuda()
    dout = (torch.randn(1, D, H, W, 3, generator=g) * 1e-3).cuda()
    L, P, S = _lib.load(), ops._p, ops._stream
    out, lse = torch.empty(1, D, H, W, 3, device="cuda"), torch.empty(1, D, H, W, 1, device="cuda")
    _lib.check(L.modet_na_fwd(P(q), P(k), P(rpb), P(out), P(lse), 1, D, H, W, 1, 6, 1.0, S()), "modet_na_fwd")
    nb = L.modet_na_bwd_ws_bytes(1, D, H, W, 1)

    def bwd(dq, dk, dr, ws):
        _lib.check(L.modet_na_bwd(P(q), P(k), P(rpb), P(out), P(lse), P(dout), P(dq), P(dk), P(dr), P(ws), nb, 1, D, H, W, 1, 6, 1.0,
                                  S()), "modet_na_bwd")
    ref = [torch.empty_like(q), torch.empty_like(k), torch.empty_like(rpb)]
    bwd(*ref, torch.empty(nb // 4 + 1, device="cuda"))
    flag = str(tmp_path / "ready")
    noise = subprocess.Popen([sys.executable, os.path.join(ROOT, "tests", "noise_worker.py"), "30", flag],
                             stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
    try:
        t0 = time.time()
        while not os.path.exists(flag) and time.time() - t0 < 120 and noise.poll() is None:
            time.sleep(0.5)
        assert os.path.exists(flag), "the noise process did not start"
        streams = [torch.cuda.Stream(priority=-1) for _ in range(4)] + [torch.cuda.Stream() for _ in range(4)]
        ticks = [torch.zeros(256, device="cuda") for _ in streams]
        got = [torch.empty_like(t) for t in ref]
        ws = torch.empty(nb // 4 + 1, device="cuda")
        bad = []
        for i in range(6000):
            bwd(*got, ws)
            bad.append(torch.stack([(a != b).sum() for a, b in zip(got, ref)]))
            if i % 4 == 0:
                for st, t in zip(streams, ticks):
                    with torch.cuda.stream(st):
                        t.add_(1.0)
        torch.cuda.synchronize()
        assert noise.poll() is None, "the noise process ended before the calls did"
    finally:
        noise.kill()
    nbad = int((torch.stack(bad).sum(1) > 0).sum())
    _note("na_bwd.calls_not_bit_identical_beside_another_process_of_6000", nbad)
    assert nbad == 0, f"{nbad} of 6000 calls differ from the first"


def test_device_volume_cache_on_the_gpu_equals_the_reference_pipeline(tmp_path):
    """SURVEY 8(f) rank 2 on the GPU box: `.pkl` subjects -> DeviceVolumeCache(device="cuda") through the
    pinned double-buffered H2D path -> device-resident pairs EQUAL to what the reference's datasets + transforms returned
    (fixture generated from /root/reference/ModeT/data/{datasets,trans}.py by tests/golden/make_goldens_data.py)."""
    from tests.util import check_data_pipeline_golden
    assert check_data_pipeline_golden(tmp_path, "cuda") == 4


@pytest.mark.parametrize("mode", ["default", "set_deterministic"])
def test_deterministic_train_step_is_bit_reproducible(mode, monkeypatch):
    """The warp scatter holds the step's only atomics.  Round 6: the DEFAULT path of every feature warp is the
    destination-tile kernel (integer sums in LDS, csrc/warp_tile.hip), so the plain train step is bit-reproducible: two eager
    steps, two trainers, and the captured hipGraph's replays all return IDENTICAL gradients -- the comparison every
    graph-vs-eager check otherwise makes through a 1e-6 noise floor.  ops.set_deterministic (round 5: 64-bit integer atomics on
    global memory for whatever the tiles do not take) must stay so.  Against the float-atomic kernel (ops.WARP_TILES off): the
    same gradients within its run-to-run noise."""
    from smilecode_amd import ops
    from smilecode_amd.engine import Trainer
    shape = (32, 48, 32)
    mov, fix = _pair(shape)
    assert ops.WARP_TILES and not ops.DETERMINISTIC, "the product defaults"
    prev = ops.set_deterministic(mode == "set_deterministic")
    try:
        a, b = Trainer(_model(shape, 1.0)), Trainer(_model(shape, 1.0))
        a._fwd_bwd(mov, fix)
        g1 = a.fp.grad.clone()
        a._fwd_bwd(mov, fix)
        assert torch.equal(a.fp.grad, g1), "two eager steps of one trainer differ"
        b._fwd_bwd(mov, fix)
        b._fwd_bwd(mov, fix)
        assert torch.equal(b.fp.grad, g1), "two trainers differ"
        b.capture(mov, fix)
        for _ in range(3):
            b.fp.grad.fill_(float("nan"))
            b._graph.replay()
            assert torch.equal(b.fp.grad, g1), "a hipGraph replay differs from the eager step"
    finally:
        ops.set_deterministic(prev)
    monkeypatch.setattr(ops, "WARP_TILES", False)
    c = Trainer(_model(shape, 1.0))
    c._fwd_bwd(mov, fix)
    gerr = float((c.fp.grad - g1).abs().max() / g1.abs().max())
    _note(f"deterministic[{mode}].grad_relerr_vs_float_atomics", gerr)
    assert gerr < 1e-5, gerr


@pytest.mark.parametrize("overlap", [False, True])
def test_step_seeded_with_the_loss_gradients_equals_loss_backward(overlap):
    """The product step starts its backward from (y_moved, flow) with the gradients the two loss kernels wrote
    (Trainer._seeded_loss: channels-last Grad3d, no planar copy of the flow, no d * 1.0 passes, no ``ones_like`` root)
    instead of from the scalar.  With the reference's weights [1, 1] (train.py:106) that is the same arithmetic element for
    element: in deterministic mode (fixed-point warp scatter) losses and gradients are IDENTICAL to
    ``Trainer.loss(...)[0].backward()``; with other weights the weight enters one multiplication earlier (<= 1e-6)."""
    from smilecode_amd import ops
    from smilecode_amd.engine import Trainer
    shape = (32, 48, 32)
    mov, fix = _pair(shape)
    prev = ops.set_deterministic(True)
    try:
        for weights in ((1.0, 1.0), (0.7, 2.5)):
            res = {}
            for seeded in (True, False):
                tr = Trainer(_model(shape, 1.0), weights=weights, overlap_allreduce=overlap)
                tr.seed_backward = seeded
                assert tr._seedable() == seeded
                out = tr._fwd_bwd_staged(mov, fix) if overlap else tr._fwd_bwd(mov, fix)
                res[seeded] = (tr.fp.grad.clone(), [float(v) for v in out])
            (ga, la), (gb, lb) = res[True], res[False]
            assert la[1] == lb[1], "NCC value"
            assert abs(la[2] - lb[2]) <= 2e-6 * abs(lb[2]) and abs(la[0] - lb[0]) <= 2e-6 * abs(lb[0]), (la, lb)
            if weights == (1.0, 1.0):
                assert torch.equal(ga, gb), float((ga - gb).abs().max())
            else:
                gerr = float((ga - gb).abs().max() / gb.abs().max())
                _note(f"seeded_step[overlap={int(overlap)}].grad_relerr_weights_0.7_2.5", gerr)
                assert gerr < 2e-6, gerr
    finally:
        ops.set_deterministic(prev)
    # a loss the kernels' value-and-gradient calls do not cover keeps the autograd path
    from smilecode_amd import losses
    tr = Trainer(_model(shape, 1.0))
    tr.sim = losses.NCC_vxm(win=[5, 3, 7])
    assert not tr._seedable()
    tr._fwd_bwd(mov, fix)
    assert bool(torch.isfinite(tr.fp.grad).all())


@pytest.mark.parametrize("bf16", [False, True])
def test_leaf_reductions_in_one_launch_equal_the_per_level_ones(bf16, monkeypatch):
    """The attention's d_rpb and the projection's d_gamma / d_beta / d_bias / d_W of ALL levels are summed by one
    modet_leaf_reduce_many launch at the end of the backward pass (15 launches of 5-8 us otherwise).  Same partial rows, fp64
    sums in another order: with the deterministic warp scatter every other gradient is IDENTICAL and these agree to the last
    bit or two; the plain and the three-stage backward both take the path."""
    from smilecode_amd import models, ops, synth
    from smilecode_amd.engine import Trainer
    shape = (32, 48, 32)
    mov, fix = _pair(shape)
    prev = ops.set_deterministic(True)
    try:
        for overlap in (False, True):
            res = []
            for defer in (True, False):
                monkeypatch.setattr(ops, "DEFER_LEAF_REDUCTIONS", defer)
                m = models.ModeT(shape, head_dim=6, num_heads=[8, 4, 2, 1, 1], scale=1.0,
                                 act_dtype=torch.bfloat16 if bf16 else torch.float32).cuda()
                models.load_numpy_weights(m, synth.make_weights(24))
                tr = Trainer(m, overlap_allreduce=overlap)
                tr.fp.grad.fill_(float("nan"))
                tr._fwd_bwd_staged(mov, fix) if overlap else tr._fwd_bwd(mov, fix)
                assert bool(torch.isfinite(tr.fp.grad).all())
                res.append((tr.fp.grad.clone(), {n: tr.fp.offsets[i] for i, (n, _) in enumerate(m.named_parameters())}))
            (ga, off), (gb, _) = res
            leaf = torch.zeros_like(ga, dtype=torch.bool)
            for n, (o, k) in off.items():
                if n.startswith(("projblock", "mdt")):
                    leaf[o:o + k] = True
            assert torch.equal(ga[~leaf], gb[~leaf]), "a gradient that is not a deferred leaf reduction changed"
            scale = float(gb[leaf].abs().max())
            err = float((ga[leaf] - gb[leaf]).abs().max())
            _note(f"leaf_reduce[bf16={int(bf16)},overlap={int(overlap)}].maxdiff_of_max", err / scale)
            assert err <= 2e-7 * scale, (err, scale)
    finally:
        ops.set_deterministic(prev)


def test_staged_graphs_follow_the_parameters():
    """the three stage graphs must pack the conv weights INSIDE graph 0 -- packed once at capture, every
    replay after the first optimizer step would convolve with the weights of capture time.  Capture, then change every
    parameter by 10 % and compare the replayed gradients with the eager staged step on the same parameters; then three
    optimizer steps through the graphs against the eager trainer."""
    from smilecode_amd.engine import Trainer
    shape = (32, 48, 32)
    mov, fix = _pair(shape)
    a, b = Trainer(_model(shape, 1.0), overlap_allreduce=True), Trainer(_model(shape, 1.0), overlap_allreduce=True)
    b.capture(mov, fix)
    assert b._stage_graphs is not None and len(b._stage_graphs) == 3
    for t in (a, b):
        t.fp.flat.mul_(1.1)
    a._fwd_bwd_staged(mov, fix)
    b.fp.grad.fill_(float("nan"))
    for gr in b._stage_graphs:
        gr.replay()
    gerr = float((a.fp.grad - b.fp.grad).abs().max() / a.fp.grad.abs().max())
    _note("staged_graph.grad_relerr_after_parameter_change", gerr)
    assert gerr < 2e-5, gerr
    for step in range(3):
        la, lb = a.train_step(mov, fix), b.train_step(mov, fix)
        assert abs(float(la[0]) - float(lb[0])) < 4e-3, (step, float(la[0]), float(lb[0]))
    assert float((a.fp.flat - b.fp.flat).abs().max()) < 1e-3


def test_checkpoint_resume_is_identical_to_never_stopping(tmp_path):
    """save {'state_dict','optimizer'} -> load into a fresh model/Trainer -> parameters, Adam m/v/vmax and the
    step count are restored BIT FOR BIT, so the next step is the step an uninterrupted run takes.  (A step itself is not
    bitwise reproducible run to run -- the feature-warp backward scatters with float atomics, as ATen's grid_sampler
    backward does -- so the continued step is compared at the level two identical resumes differ from each other.)"""
    from smilecode_amd import models
    from smilecode_amd.engine import Trainer
    shape = (32, 48, 32)
    mov, fix = _pair(shape)
    b = Trainer(_model(shape, 1.0))
    for _ in range(2):
        b.train_step(mov, fix, epoch=1)
    path = str(tmp_path / "ck.pth.tar")
    torch.save({"epoch": 2, "state_dict": b.model.state_dict(), "best_dsc": 0.5, "optimizer": b.state_dict()}, path)
    ck = torch.load(path, map_location="cpu")
    assert set(ck["optimizer"]) == {"state", "param_groups"} and len(ck["optimizer"]["state"]) == len(list(b.model.parameters()))

    def resumed(with_optimizer=True):
        m = models.ModeT(shape, head_dim=6, num_heads=[8, 4, 2, 1, 1], scale=1).cuda()
        m.load_state_dict(ck["state_dict"])
        t = Trainer(m)
        if with_optimizer:
            t.load_state_dict(ck["optimizer"])
        return t

    c1, c2, d = resumed(), resumed(), resumed(False)
    assert c1.step == 2 and torch.equal(c1.fp.flat, b.fp.flat)
    assert torch.equal(c1.m, b.m) and torch.equal(c1.v, b.v) and torch.equal(c1.vmax, b.vmax), "Adam state not restored bit for bit"
    for t in (b, c1, c2, d):
        t.train_step(mov, fix, epoch=1)
    # parameters with a real gradient (Adam turns the pure-noise gradient of a conv bias in front of an InstanceNorm --
    # analytically zero -- into a +-lr step of arbitrary sign)
    sig = torch.zeros_like(b.fp.flat, dtype=torch.bool)
    for (n, p), (off, k) in zip(b.model.named_parameters(), b.fp.offsets):
        if not (n.endswith("main.bias") and "conv.2" not in n and "conv0.0" not in n):
            sig[off:off + k] = True

    def mean_diff(x, y):
        return float((x.fp.flat - y.fp.flat)[sig].abs().mean())

    noise, resume_err, restart_err = mean_diff(c1, c2), mean_diff(c1, b), mean_diff(d, b)
    _note("resume.mean_param_diff_between_two_resumes", noise)
    _note("resume.mean_param_diff_vs_uninterrupted", resume_err)
    _note("resume.mean_param_diff_without_optimizer_state", restart_err)
    assert resume_err <= 3.0 * noise + 1e-8, f"resumed step differs from the uninterrupted one: {resume_err:.2e} (run-to-run {noise:.2e})"
    # without the optimizer state (the reference's own resume, train.py:80-85) Adam restarts: the step differs by ~lr
    assert restart_err > 1e-5 and restart_err > 30.0 * resume_err


def test_hip_graph_step_equals_eager_step():
    """forward+backward captured into one hipGraph (Trainer.capture) gives the eager step's loss and
    gradients, replays on new inputs, and costs the host < 1 ms per step"""
    import time
    from smilecode_amd.engine import Trainer
    shape = (32, 48, 32)
    mov, fix = _pair(shape)
    a, b = Trainer(_model(shape, 1.0)), Trainer(_model(shape, 1.0))
    b.capture(mov, fix)
    assert torch.equal(a.fp.flat, b.fp.flat), "capture (and its warm-up) must not touch the parameters"
    la, lb = a.train_step(mov, fix), b.train_step(mov, fix)
    assert abs(float(la[0]) - float(lb[0])) < 1e-6 and abs(float(la[2]) - float(lb[2])) < 1e-6
    gerr = float((a.fp.grad - b.fp.grad).abs().max() / a.fp.grad.abs().max())
    assert gerr < 1e-4, gerr
    _note("graph.grad_relerr_vs_eager", gerr)
    # new inputs through the same graph (static input buffers are refreshed per step)
    from smilecode_amd import synth
    m2, f2 = (torch.from_numpy(t).cuda() for t in synth.make_pair(shape, 31))
    c = Trainer(_model(shape, 1.0))
    lc = c.train_step(m2, f2)
    b2 = Trainer(_model(shape, 1.0)).capture(mov, fix)
    lb2 = b2.train_step(m2, f2)
    assert abs(float(lc[0]) - float(lb2[0])) < 1e-6
    # another shape falls back to the eager HIP path
    m3, f3 = _pair((32, 32, 32))
    b3 = Trainer(_model((32, 32, 32), 1.0))
    b3._graph, b3._graph_key = b2._graph, b2._graph_key
    assert torch.isfinite(b3.train_step(m3, f3)[0])
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(10):
        b.train_step(mov, fix)
    host = (time.perf_counter() - t0) / 10 * 1e3
    torch.cuda.synchronize()
    _note("graph.host_enqueue_ms_per_step", host)
    assert host < 1.0, f"graph replay should cost the host well under 1 ms per step, took {host:.2f}"


def test_hip_graph_training_trajectory_equals_eager():
    """EVERY replay must be the eager step, not only the first: 12 Adam steps through the replayed graph follow the eager
    run's losses and end at the same parameters.  (Round 3 found the captured step wrong from its SECOND replay on: a
    hipMemsetAsync captured as a memset node only clears its buffer on the first replay on this ROCm, so the scatter-add of
    the warp backward accumulated onto garbage -- gradients of 1e20+, NaN parameters within ~15 steps in a third of the
    runs.  The one-step test above could not see it.  csrc/common.h: modet_zero_async.)"""
    from smilecode_amd.engine import Trainer
    shape = (32, 48, 32)
    mov, fix = _pair(shape)
    a, b = Trainer(_model(shape, 1.0)), Trainer(_model(shape, 1.0))
    b.capture(mov, fix)
    first = None
    for step in range(12):
        la, lb = a.train_step(mov, fix), b.train_step(mov, fix)
        first = float(la[0]) if first is None else first
        gb = b.fp.grad
        assert bool(torch.isfinite(gb).all()) and float(gb.abs().max()) < 1e3, f"gradient of the replayed graph at step {step}"
        # Adam's update has unit scale whatever the gradient's size, so the float-atomic reordering of the warp backward
        # separates two runs of the SAME code by a few per cent of a gradient within a handful of steps: the losses are
        # the stable quantity
        assert abs(float(la[0]) - float(lb[0])) < 4e-3, (step, float(la[0]), float(lb[0]))
    assert float(lb[0]) < first - 0.02 and bool(torch.isfinite(b.fp.flat).all()), (first, float(lb[0]))
    perr = float((a.fp.flat - b.fp.flat).abs().max())
    assert perr < 2.5e-3, perr                   # 12 steps of lr 1e-4: no parameter moved by more than 1.2e-3
    _note("graph.loss_diff_vs_eager_after_12_steps", abs(float(la[0]) - float(lb[0])))
    # the same replay, gradients only, three times on fixed parameters: identical to eager each time
    c = Trainer(_model(shape, 1.0))
    c._fwd_bwd(mov, fix)
    ref = c.fp.grad.clone()
    c.capture(mov, fix)
    for rep in range(3):
        c.fp.grad.fill_(float("nan"))
        c._graph.replay()
        torch.cuda.synchronize()
        assert float((c.fp.grad - ref).abs().max() / ref.abs().max()) < 1e-4, rep


def test_graph_survives_an_eager_step_of_another_shape():
    """the captured graph bakes in the address of the packed-weights arena.  An eager step of ANOTHER shape on
    the same Trainer (train_step's key-mismatch fallback) must leave that arena alone: replaying the first shape
    afterwards still gives the eager gradients (each shape has its own ops.StepContext, alive as long as the Trainer)."""
    from smilecode_amd.engine import Trainer
    sa, sb = (32, 48, 32), (32, 32, 48)
    mov, fix = _pair(sa)
    mb, fb = _pair(sb)
    ref = Trainer(_model(sa, 1.0))
    t = Trainer(_model(sa, 1.0)).capture(mov, fix)
    arena = [sc.arena.data_ptr() for sc in t._steps.values()]
    assert len(arena) == 1
    l_ref = ref._fwd_bwd(mov, fix)
    g_ref = ref.fp.grad.clone()
    # an eager pass at another shape (same parameters: the encoder is fully convolutional), twice: record + batched
    for _ in range(2):
        lb = t._fwd_bwd(mb, fb)
        assert torch.isfinite(lb[0])
    junk = [torch.full((1 << 22,), float("nan"), device="cuda") for _ in range(4)]    # anything freed would be re-used now
    assert len(t._steps) == 2 and arena[0] in [sc.arena.data_ptr() for sc in t._steps.values()]
    t._static_in[0].copy_(mov); t._static_in[1].copy_(fix)
    t._graph.replay()
    torch.cuda.synchronize()
    del junk
    assert abs(float(t._static_out[0]) - float(l_ref[0])) < 1e-6
    gerr = float((t.fp.grad - g_ref).abs().max() / g_ref.abs().max())
    _note("graph.grad_relerr_after_shape_switch", gerr)
    assert gerr < 1e-4, gerr


def test_two_trainers_on_two_threads_equal_running_alone():
    """SURVEY 8(b) "re-entrant, no global mutable state" / VERDICT r2 item 7: the step-batching tables live in caller-owned
    contexts (modet_step_ctx_t), so two Trainers stepping concurrently from two threads of one process -- different
    weights, different inputs, own streams -- produce the gradients they produce alone."""
    import threading
    from smilecode_amd import models, synth
    from smilecode_amd.engine import Trainer
    shape = (32, 48, 32)

    def make(seed):
        m = models.ModeT(shape, head_dim=6, num_heads=[8, 4, 2, 1, 1], scale=1).cuda()
        models.load_numpy_weights(m, synth.make_weights(seed))
        mov, fix = synth.make_pair(shape, seed)
        return Trainer(m), torch.from_numpy(mov).cuda(), torch.from_numpy(fix).cuda()

    alone = []
    for seed in (24, 77):
        t, mov, fix = make(seed)
        for _ in range(2):                                   # recording pass, then the batched pass
            t._fwd_bwd(mov, fix)
        alone.append(t.fp.grad.clone())
    assert not torch.allclose(alone[0], alone[1])
    pairs = [make(24), make(77)]
    torch.cuda.synchronize()
    errs, barrier = [None, None], threading.Barrier(2)

    def work(i):
        try:
            t, mov, fix = pairs[i]
            st = torch.cuda.Stream()
            with torch.cuda.stream(st):
                for k in range(4):
                    barrier.wait(timeout=120)                # both threads inside a step at the same time, every time
                    t._fwd_bwd(mov, fix)
            st.synchronize()
            errs[i] = float((t.fp.grad - alone[i]).abs().max() / alone[i].abs().max())
        except Exception as e:                               # noqa: BLE001
            errs[i] = e
            barrier.abort()

    th = [threading.Thread(target=work, args=(i,)) for i in range(2)]
    for x in th:
        x.start()
    for x in th:
        x.join(300)
    for e in errs:
        assert isinstance(e, float), e
        assert e < 1e-4, errs                                # the feature-warp scatter's float atomics: run-to-run ~1e-6
    _note("two_trainers_two_threads.grad_relerr", max(errs))


def test_reference_train_loop_call_order(tmp_path):
    """the reference's own loop body (train.py:122-133) on our modules: losses are called as
    loss_function(output[n], y), i.e. the tensor that needs gradients is NCC's FIRST argument"""
    from smilecode_amd import losses
    shape = (32, 48, 32)
    model = _model(shape, 1.0)
    x, y = _pair(shape)
    criterions = [losses.NCC_vxm(), losses.Grad3d(penalty="l2")]
    weights = [1, 1]
    optimizer = torch.optim.Adam(model.parameters(), lr=1e-4, weight_decay=0, amsgrad=True)
    output = model(x, y)
    loss = 0
    for n, loss_function in enumerate(criterions):
        loss = loss + loss_function(output[n], y) * weights[n]
    optimizer.zero_grad()
    loss.backward()
    optimizer.step()
    g = gold("e2e_32x48x32.npz")
    assert abs(float(loss) - float(g["loss"][0])) < 2e-4
    ref = g["grad.mdt1.rpb"]
    got = np64(model.mdt1.rpb.grad).reshape(-1)
    assert np.abs(got - ref.reshape(-1)).max() <= 2e-2 * np.abs(ref).max()


def test_train_and_infer_scripts_synthetic(tmp_path):
    """the train.py / infer.py equivalents run end to end (synthetic subjects, 64^3, 2 iterations), write the
    reference's checkpoint dict and log files, and the checkpoint loads back through infer."""
    import glob as _glob
    import sys
    from smilecode_amd import infer, train
    out = str(tmp_path)
    stdout = sys.stdout
    try:
        train.main(["--synthetic", "3", "--img-size", "64,64,64", "--max-epoch", "1", "--max-iters", "2", "--out", out])
    finally:
        sys.stdout = stdout
    ck = _glob.glob(out + "/experiments/*/dsc*.pth.tar")
    assert len(ck) == 1
    sd = torch.load(ck[0], map_location="cpu")
    assert set(sd) == {"epoch", "state_dict", "best_dsc", "optimizer"} and "encoder.conv0.0.main.weight" in sd["state_dict"]
    assert set(sd["optimizer"]) == {"state", "param_groups"} and float(sd["optimizer"]["state"][0]["step"]) == 2.0
    # resume (train.py:59-62,:80-85): --cont-training --epoch-start 1 picks the checkpoint up and runs epoch 1
    try:
        train.main(["--synthetic", "3", "--img-size", "64,64,64", "--max-epoch", "2", "--max-iters", "1", "--out", out,
                    "--cont-training", "--epoch-start", "1"])
    finally:
        sys.stdout = stdout
    cks = [torch.load(c, map_location="cpu") for c in _glob.glob(out + "/experiments/*/dsc*.pth.tar")]
    assert max(float(c["optimizer"]["state"][0]["step"]) for c in cks) == 3.0 and max(c["epoch"] for c in cks) == 2
    log = open(_glob.glob(out + "/logs/*/logfile.log")[0]).read()
    assert "Iter 1 of 2 loss" in log and "Img Sim:" in log
    d = infer.main(["--synthetic", "2", "--img-size", "64,64,64", "--model-dir", os.path.dirname(ck[0]) + "/"])
    assert 0.0 <= d <= 1.0
