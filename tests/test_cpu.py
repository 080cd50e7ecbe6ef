"""CPU suite (-m "not gpu"): the oracle against the golden vectors captured from the reference, the host
logic, and that libmodet_hip.so loads and exports every symbol include/modet_hip.h declares (no compute
calls: there is no GPU here)."""
import ctypes
import os
import subprocess
import sys

import numpy as np
import pytest
import torch

from tests.util import assert_close, gold

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HEADS = (8, 4, 2, 1, 1)


@pytest.fixture(scope="module")
def orc():
    from oracle import modet_torch
    return modet_torch


def T(a):
    return torch.from_numpy(np.ascontiguousarray(a)).double()


# ------------------------------------------------------------------------------------------------ library
def test_library_builds_loads_and_exports_header_symbols():
    from smilecode_amd import _lib, build
    path = build.build(verbose=False)
    assert os.path.exists(path)
    lib = _lib.load()
    declared = _lib.header_symbols()
    assert len(declared) >= 35
    for name in declared:
        assert hasattr(lib, name), f"{name} declared in include/modet_hip.h but not exported"
        assert name in _lib.SIGNATURES, f"{name} has no ctypes signature"
    assert set(_lib.SIGNATURES) == set(declared)
    assert lib.modet_hip_version() >= 100
    assert _lib.strerror(0) == "ok" and "NULL" in _lib.strerror(-1)
    # pure host entry points (no device needed)
    assert lib.modet_na_bwd_ws_bytes(1, 160, 192, 160, 1) >= 10 * 24 * 80 * 27 * 4
    assert lib.modet_conv3d_ws_bytes(8, 8) >= 27 * 8 * 16 * 4


def test_product_path_has_no_cpu_fallback():
    from smilecode_amd import ops
    x = torch.zeros(1, 4, 4, 4, 8)
    with pytest.raises(RuntimeError, match="GPU"):
        ops.instnorm_lrelu(x)
    with pytest.raises(RuntimeError, match="GPU"):
        ops.warp(x, torch.zeros(1, 4, 4, 4, 3))
    # the product package never imports the oracle
    src = subprocess.run(["grep", "-rl", "oracle", os.path.join(ROOT, "smilecode_amd"), "--include=*.py"],
                         capture_output=True, text=True).stdout.split()
    assert src == [], f"product files mention the oracle: {src}"


def test_param_spec_and_state_dict_keys():
    from smilecode_amd import models, synth
    spec = synth.param_spec()
    m = models.ModeT((32, 48, 32))
    assert [n for n, _ in m.named_parameters()] == list(spec.keys())
    for n, p in m.named_parameters():
        assert tuple(p.shape) == spec[n]
    assert sum(p.numel() for p in m.parameters()) == 1029670
    sd = m.state_dict()
    assert tuple(sd["mdt3.grid"].shape) == (3, 3, 3, 3)
    assert not any(k.startswith("transformer.") for k in sd)
    legacy = models.ModeT((32, 48, 32), legacy_grid_buffers=True).state_dict()
    assert tuple(legacy["transformer.1.grid"].shape) == (1, 3, 16, 24, 16)
    assert float(legacy["transformer.0.grid"][0, 1, 3, 5, 7]) == 5.0
    cu = models.ModeT_cu((32, 48, 32))
    assert tuple(cu.state_dict()["mdt1.v"].shape) == (27, 3) and cu.mdt1.scale == 1
    cu.load_state_dict(legacy, strict=True)                          # grid flavour + transformer grids accepted
    with pytest.raises(RuntimeError):
        models.ModeT((30, 48, 32))


def test_synth_is_deterministic():
    from smilecode_amd import synth
    a, b = synth.make_pair((16, 16, 16), 24)
    a2, _ = synth.make_pair((16, 16, 16), 24)
    assert np.array_equal(a, a2) and a.min() == 0.0 and a.max() <= 1.0 and not np.array_equal(a, b)
    lab = synth.make_labels((32, 48, 32), 24)
    u = set(int(v) for v in np.unique(lab))
    assert lab.dtype == np.int16 and u <= set(range(55)) and len(u) >= 50
    w = synth.make_weights(24)
    assert abs(float(w["mdt1.rpb"].std()) - 0.5) < 0.2


# ------------------------------------------------------------------------------------------------ oracle pins
def test_oracle_attention_golden(orc):
    g = gold("op_attention.npz")
    for tag in ("h1", "h2", "h8", "h4"):
        heads = int(tag[1:])
        q, k, rpb = T(g[f"{tag}.q"]).requires_grad_(True), T(g[f"{tag}.k"]).requires_grad_(True), T(g[f"{tag}.rpb"]).requires_grad_(True)
        out = orc.mode_transformer(q, k, rpb, heads, float(g[f"{tag}.scale"]))
        assert_close(out.detach().numpy(), g[f"{tag}.out"], atol=1e-12, rtol=0, what="attention out")
        dq, dk, dr = torch.autograd.grad(out, [q, k, rpb], T(g[f"{tag}.gy"]))
        assert_close(dq.numpy(), g[f"{tag}.dq"], atol=1e-12, rtol=0)
        assert_close(dk.numpy(), g[f"{tag}.dk"], atol=1e-12, rtol=0)
        assert_close(dr.numpy(), g[f"{tag}.drpb"], atol=1e-11, rtol=0)


def test_oracle_warp_golden(orc):
    g = gold("op_warp.npz")
    for tag in ("a", "b", "c"):
        src, flow = T(g[f"{tag}.src"]).requires_grad_(True), T(g[f"{tag}.flow"]).requires_grad_(True)
        out = orc.warp(src, flow)
        assert_close(out.detach().numpy(), g[f"{tag}.out"], atol=1e-12, rtol=0)
        ds, df = torch.autograd.grad(out, [src, flow], T(g[f"{tag}.gy"]))
        assert_close(ds.numpy(), g[f"{tag}.dsrc"], atol=1e-12, rtol=0)
        assert_close(df.numpy(), g[f"{tag}.dflow"], atol=1e-11, rtol=0)
        assert np.array_equal(orc.warp(T(g[f"{tag}.lab"]), T(g[f"{tag}.flow_n"]), "nearest").numpy(), g[f"{tag}.out_n"])


def test_oracle_losses_and_dice_golden(orc):
    from smilecode_amd import synth
    g = gold("op_misc.npz")
    a, b = T(g["ncc.a"]), T(g["ncc.b"]).requires_grad_(True)
    l = orc.ncc_loss(a, b)
    assert abs(float(l) - float(g["ncc.val"])) < 1e-12
    assert_close(torch.autograd.grad(l, b)[0].numpy(), g["ncc.db"], atol=1e-13, rtol=0)
    f = T(g["g3d.flow"])
    assert abs(float(orc.grad3d_loss(f)) - float(g["g3d.val"])) < 1e-12
    d = gold("op_dice.npz")
    shape = tuple(int(s) for s in d["shape"])
    lm, lf = T(synth.make_labels(shape, 24))[None, None], T(synth.make_labels(shape, 25))[None, None]
    w = orc.warp(lm, T(d["flow"]), "nearest")
    assert np.array_equal(w[0, 0].numpy().astype(np.int16), d["warped"])
    assert abs(orc.dice_voi(w.long(), lf.long()) - float(d["dice"])) < 1e-12


def test_oracle_end_to_end_golden(orc):
    from smilecode_amd import synth
    g = gold("e2e_32x48x32.npz")
    shape = (32, 48, 32)
    p = {n: T(v).requires_grad_(True) for n, v in synth.make_weights(24).items()}
    mov, fix = (T(a) for a in synth.make_pair(shape, 24))
    loss, sim, reg, y, flow = orc.train_loss(p, mov, fix, HEADS, 6, 1.0)
    assert_close(flow.detach().numpy().reshape(-1), g["flow"], atol=2e-6, rtol=0, what="flow (fixture is fp32-rounded)")
    assert_close(y.detach().numpy().reshape(-1), g["y_moved"], atol=2e-7, rtol=0)
    assert_close(np.array([float(loss), float(sim), float(reg)]), g["loss"], atol=1e-12, rtol=0)
    names = ["mdt1.rpb", "projblock3.proj.weight", "encoder.conv0.0.main.weight", "cwm5.conv.2.bias", "encoder.conv4.2.main.weight"]
    grads = torch.autograd.grad(loss, [p[n] for n in names])
    for n, gr in zip(names, grads):
        ref = g["grad." + n]
        got = gr.numpy().reshape(-1)
        got = got if got.size == ref.size else got[::61]
        assert_close(got, ref.reshape(-1), atol=1e-12, rtol=1e-9, what=n)


def test_oracle_adam_matches_torch_optim(orc):
    gen = torch.Generator().manual_seed(0)
    w = torch.randn(50, generator=gen).double()
    p = torch.nn.Parameter(w.clone())
    opt = torch.optim.Adam([p], lr=1e-3, amsgrad=True)
    pr, st = {"w": w.clone()}, {"w": (torch.zeros(50).double(), torch.zeros(50).double(), torch.zeros(50).double())}
    for step in range(1, 4):
        gr = torch.randn(50, generator=gen).double()
        p.grad = gr.clone()
        opt.step()
        orc.adam_amsgrad_step(pr, {"w": gr}, st, 1e-3, step)
    assert_close(pr["w"].numpy(), p.detach().numpy(), atol=1e-14, rtol=0)
    assert orc.poly_lr(0) == 1e-4 and orc.poly_lr(15) == round(1e-4 * 0.5 ** 0.9, 8)


# ------------------------------------------------------------------------------------------------ host logic
def test_poly_lr_and_pair_sharding():
    from smilecode_amd.engine import poly_lr
    from smilecode_amd.parallel import pairs_for_rank
    assert poly_lr(0) == 1e-4 and poly_lr(29) == round(1e-4 * (1 - 29 / 30) ** 0.9, 8)
    got = sorted(sum((pairs_for_rank(19, r, 4) for r in range(4)), []))
    assert got == list(range(19))


def _dp_worker(rank, world, port, q):
    import torch.distributed as dist
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank))
    from smilecode_amd.parallel import FlatParams, broadcast_parameters, init_from_env
    init_from_env("gloo")
    torch.manual_seed(rank)                       # ranks start different; broadcast must align them
    net = torch.nn.Sequential(torch.nn.Linear(5, 7), torch.nn.Linear(7, 3))
    fp = FlatParams(net)
    broadcast_parameters(fp)
    x = torch.full((4, 5), float(rank + 1))
    fp.zero_grad()
    net(x).sum().backward()
    fp.gather_grads()                                                 # one fused copy into the flat buffer
    assert torch.equal(fp.grad[:35].view(7, 5), net[0].weight.grad)
    local = fp.grad.clone()
    scale = fp.allreduce_grads()
    q.put((rank, fp.flat.clone().numpy(), local.numpy(), (fp.grad * scale).numpy()))
    dist.destroy_process_group()


def test_data_parallel_allreduce_gloo_world2():
    """N>1 path on CPU: flat buffers, parameter broadcast, one all-reduce, 1/world scale."""
    import torch.multiprocessing as mp
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 29611 + os.getpid() % 200
    procs = [ctx.Process(target=_dp_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = sorted([q.get(timeout=120) for _ in procs], key=lambda t: t[0])
    for p in procs:
        p.join(60)
        assert p.exitcode == 0
    assert np.array_equal(res[0][1], res[1][1]), "parameters must be identical after the broadcast"
    mean = 0.5 * (res[0][2] + res[1][2])
    assert np.allclose(res[0][3], mean) and np.allclose(res[1][3], mean)
    assert not np.allclose(res[0][2], res[1][2])


def test_data_pipeline_pairs_and_label_remap(tmp_path):
    """all-ordered-pairs indexing and Seg_norm table of the reference (datasets.py:24-26, trans.py:27-39)"""
    import pickle
    from smilecode_amd import data
    n = 4
    seen = [data.pair_indices(i, n) for i in range(n * (n - 1))]
    assert len(set(seen)) == n * (n - 1) and all(a != b for a, b in seen)
    assert seen[:4] == [(0, 1), (0, 2), (0, 3), (1, 0)]
    lab = np.array([[0, 21, 34], [41, 166, 999], [20, 122, 161]], dtype=np.uint16)
    assert data.seg_norm(lab).tolist() == [[0, 1, 14], [15, 54, 0], [0, 48, 49]]
    for i in range(3):
        with open(tmp_path / f"s{i}.pkl", "wb") as f:
            pickle.dump((np.full((4, 4, 4), i, np.float32), np.full((4, 4, 4), 21 + i, np.uint16)), f)
    ds = data.LPBABrainInferDatasetS2S([str(p) for p in tmp_path.glob("*.pkl")])
    assert len(ds) == 6
    x, y, xs, ys = ds[2]                                   # pair (1, 0) for n = 3
    assert x.shape == (1, 4, 4, 4) and float(x[0, 0, 0, 0]) == 1.0 and float(y[0, 0, 0, 0]) == 0.0
    assert xs.dtype == torch.int16 and int(xs[0, 0, 0, 0]) == 2 and int(ys[0, 0, 0, 0]) == 1


def test_jacobian_determinant_identity_and_fold():
    from smilecode_amd.utils import jacobian_determinant_vxm
    disp = np.zeros((3, 6, 7, 8), np.float32)
    assert np.allclose(jacobian_determinant_vxm(disp), 1.0)
    disp[0] = -2.0 * np.arange(6)[:, None, None]            # x -> -x: folding
    assert (jacobian_determinant_vxm(disp) < 0).all()


# ------------------------------------------------------------------------------------------------ C oracle pins
def test_c_oracle_against_reference_goldens(orc):
    """oracle/modet_ref.c (plain C, fp64) vs the vectors captured from the real reference, and vs the ATen oracle."""
    from oracle import cref
    from smilecode_amd import synth
    import torch.nn.functional as F
    g = gold("op_attention.npz")
    for tag in ("h1", "h2", "h8"):
        heads = int(tag[1:])
        sc = float(g[f"{tag}.scale"])
        q0, k0, rpb = g[f"{tag}.q"], g[f"{tag}.k"], g[f"{tag}.rpb"]
        B, D, H, W, Cc = q0.shape
        d = Cc // heads
        # fused form (ModeT/models.py:308-334)
        assert_close(np.moveaxis(cref.na_fwd(q0, k0, rpb.reshape(heads, 27), heads, sc), -1, 1), g[f"{tag}.out"],
                     atol=1e-12, rtol=0, what="C na_fwd")
        # CUDA-operator contract (modet_kernel.cu): logits and the three gradients
        q = np.ascontiguousarray(np.transpose(q0.reshape(B, D, H, W, heads, d), (0, 4, 1, 2, 3, 5)) * sc)
        kp = F.pad(T(k0).permute(0, 4, 1, 2, 3), (1, 1, 1, 1, 1, 1)).reshape(B, heads, d, D + 2, H + 2, W + 2)
        kp = kp.permute(0, 1, 3, 4, 5, 2).contiguous().numpy()
        attn = cref.modet_fw(q, kp, rpb.reshape(heads, 27))
        assert_close(attn, g[f"{tag}.logits"], atol=1e-12, rtol=0, what="C modet_fw")
        ga = np.random.default_rng(3).normal(size=attn.shape)
        dq, dk, dr = cref.modet_bw(ga, q, kp)
        qt, kt, rt = T(q).requires_grad_(True), T(kp).requires_grad_(True), T(rpb).requires_grad_(True)
        cols = [(qt * kt[:, :, a:a + D, b:b + H, c:c + W]).sum(-1) for a in range(3) for b in range(3) for c in range(3)]
        ref = torch.stack(cols, -1) + rt.reshape(1, heads, 1, 1, 1, 27)
        rq, rk, rr = torch.autograd.grad(ref, [qt, kt, rt], T(ga))
        assert_close(dq, rq.numpy(), atol=1e-12, rtol=0); assert_close(dk, rk.numpy(), atol=1e-12, rtol=0)
        assert_close(dr, rr.numpy(), atol=1e-11, rtol=0)
    g = gold("op_warp.npz")
    for tag in ("a", "b", "c"):
        assert_close(cref.warp(g[f"{tag}.src"], g[f"{tag}.flow"], 0), g[f"{tag}.out"], atol=1e-12, rtol=0, what="C warp")
        assert np.array_equal(cref.warp(g[f"{tag}.lab"], g[f"{tag}.flow_n"], 1), g[f"{tag}.out_n"])
    g = gold("op_misc.npz")
    for tag in ("c0", "c1", "c2"):
        raw, out = cref.conv_block(g[f"{tag}.x"], g[f"{tag}.w"], g[f"{tag}.b"], bool(g[f"{tag}.ins"]))
        assert_close(raw, g[f"{tag}.raw"], atol=1e-12, rtol=0, what="C conv")
        assert_close(out, g[f"{tag}.out"], atol=1e-11, rtol=0, what="C conv block")
    assert abs(cref.ncc(g["ncc.a"], g["ncc.b"]) - float(g["ncc.val"])) < 1e-12
    assert abs(cref.grad3d(g["g3d.flow"]) - float(g["g3d.val"])) < 1e-12
    d = gold("op_dice.npz")
    shape = tuple(int(s) for s in d["shape"])
    assert abs(cref.dice(d["warped"], synth.make_labels(shape, 25)) - float(d["dice"])) < 1e-12
