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
    assert lib.modet_hip_version() >= 200
    assert _lib.strerror(0) == "ok" and "NULL" in _lib.strerror(-1)
    # pure host entry points (no device needed)
    assert lib.modet_na_bwd_ws_bytes(1, 160, 192, 160, 1) >= 10 * 24 * 80 * 27 * 4
    assert lib.modet_conv3d_ws_bytes(8, 8) >= 27 * 8 * 16 * 4


def test_header_is_plain_c_and_the_ctypes_structs_match_it(tmp_path):
    """include/modet_hip.h is the C ABI a cgo / JNI / ctypes binding compiles against: it must parse as C (gcc -std=c99, no HIP
    headers), and the one struct that crosses the boundary by value (modet_leaf_job_t) must have the layout smilecode_amd._lib
    declares for ctypes"""
    import ctypes
    from smilecode_amd import _lib
    src = tmp_path / "abi.c"
    src.write_text(
        '#include <stdio.h>\n#include <stddef.h>\n#include "modet_hip.h"\n'
        'int main(void) { printf("%zu %zu %zu %zu %zu %zu %zu %d\\n", sizeof(modet_leaf_job_t), offsetof(modet_leaf_job_t, outer), '
        'offsetof(modet_leaf_job_t, col_group_stride), offsetof(modet_leaf_job_t, ncols), offsetof(modet_leaf_job_t, col_group), '
        'offsetof(modet_leaf_job_t, dst), offsetof(modet_leaf_job_t, n), MODET_LEAF_MAX_JOBS); return 0; }\n')
    exe = tmp_path / "abi"
    r = subprocess.run(["gcc", "-std=c99", "-Wall", "-Werror", "-I", os.path.join(ROOT, "include"), str(src), "-o", str(exe)],
                       capture_output=True, text=True)
    assert r.returncode == 0, r.stderr
    got = [int(v) for v in subprocess.run([str(exe)], capture_output=True, text=True).stdout.split()]
    J = _lib.LeafJob
    assert got[:7] == [ctypes.sizeof(J), J.outer.offset, J.col_group_stride.offset, J.ncols.offset, J.col_group.offset,
                       J.dst.offset, J.n.offset], got
    assert got[7] == 16


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
    models.ModeT((30, 48, 32))           # like the reference's constructor: accepted; forward() raises (tests/test_gpu_e2e.py)
    with pytest.raises(RuntimeError):
        models.ModeT((30, 48))


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


def _lockstep_worker(rank, world, port, q):
    import torch.distributed as dist
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank))
    from smilecode_amd.parallel import FlatParams, broadcast_parameters, init_from_env, lockstep_pairs_for_rank
    init_from_env("gloo")
    torch.manual_seed(3)
    net = torch.nn.Linear(4, 2)
    fp = FlatParams(net)
    broadcast_parameters(fp)
    n_pairs, steps = 5, 0                                   # 5 % 2 != 0: the ADVICE r1 case (870 pairs over 8 ranks)
    samples = torch.arange(n_pairs * 4, dtype=torch.float32).view(n_pairs, 4)
    for epoch in range(2):
        order = np.random.RandomState(24 + epoch).permutation(n_pairs)
        for i in lockstep_pairs_for_rank(n_pairs, rank, world):
            fp.zero_grad()
            net(samples[int(order[i])][None]).sum().backward()
            fp.gather_grads()
            scale = fp.allreduce_grads()                     # blocking: every rank must arrive the same number of times
            with torch.no_grad():
                fp.flat.add_(fp.grad, alpha=-0.01 * (epoch + 1) * scale)
            steps += 1
        dist.barrier()
    q.put((rank, steps, fp.flat.clone().numpy()))
    dist.destroy_process_group()


def test_data_parallel_lockstep_uneven_pairs_gloo_world2():
    """n_pairs % world != 0 over several steps and epochs: every rank runs the same number of all-reduces (wrap-around
    padding), nobody hangs, replicas stay bit-identical (train.py gave ranks 109 vs 108 steps)"""
    import torch.multiprocessing as mp
    from smilecode_amd.parallel import lockstep_pairs_for_rank
    for n, w in ((5, 2), (870, 8), (7, 8), (16, 8)):
        shards = [lockstep_pairs_for_rank(n, r, w) for r in range(w)]
        assert len({len(sh) for sh in shards}) == 1 and len(shards[0]) == -(-n // w)
        assert set(sum(shards, [])) == set(range(n))                     # every pair is visited
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 29411 + os.getpid() % 200
    procs = [ctx.Process(target=_lockstep_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = sorted([q.get(timeout=120) for _ in procs], key=lambda t: t[0])
    for p in procs:
        p.join(60)
        assert p.exitcode == 0
    assert res[0][1] == res[1][1] == 6
    assert np.array_equal(res[0][2], res[1][2])


def _bucket_worker(rank, world, port, q):
    import torch.distributed as dist
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank))
    from smilecode_amd.parallel import BucketedAllReduce, FlatParams, broadcast_parameters, init_from_env
    init_from_env("gloo")
    torch.manual_seed(5)
    net = torch.nn.Sequential(torch.nn.Linear(6, 8), torch.nn.Tanh(), torch.nn.Linear(8, 8), torch.nn.Tanh(), torch.nn.Linear(8, 3),
                              torch.nn.Linear(3, 3))            # module 5 is never used: its bucket must still be reduced
    fp = FlatParams(net)
    broadcast_parameters(fp)
    bk = BucketedAllReduce(fp, list(net.named_parameters()), (("4.", "5."), ("2.",), ("0.",)))
    fired = []
    orig = bk._fire
    bk._fire = lambda k: (fired.append(k), orig(k))[1]
    x = torch.full((5, 6), 0.1 * (rank + 1))
    out = []
    for it in range(2):                                         # two steps: the hooks re-arm
        fp.zero_grad()
        loss = net[4](net[3](net[2](net[1](net[0](x))))).square().sum() * (it + 1)
        fired.clear()
        bk.begin()
        loss.backward()
        order_in_backward = list(fired)
        scale = bk.finish()
        out.append((fp.grad * scale).clone().numpy())
        # reference: plain path on the same local gradient
        fp.zero_grad()
        loss = net[4](net[3](net[2](net[1](net[0](x))))).square().sum() * (it + 1)
        loss.backward()
        fp.gather_grads()
        s2 = fp.allreduce_grads()
        out.append((fp.grad * s2).clone().numpy())
    q.put((rank, out, order_in_backward, list(fired)))
    dist.destroy_process_group()


def test_bucketed_overlapped_allreduce_gloo_world2():
    """cfg 5's overlapped all-reduce: buckets fire from backward hooks in readiness order (last layers first), the result
    equals the single all-reduce after backward, unused parameters do not hang the step"""
    import torch.multiprocessing as mp
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 29211 + os.getpid() % 200
    procs = [ctx.Process(target=_bucket_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = sorted([q.get(timeout=120) for _ in procs], key=lambda t: t[0])
    for p in procs:
        p.join(60)
        assert p.exitcode == 0
    for rank, out, in_bwd, all_fired in res:
        assert np.allclose(out[0], out[1], rtol=0, atol=1e-7) and np.allclose(out[2], out[3], rtol=0, atol=1e-7)
        assert in_bwd == [1, 2], in_bwd          # buckets 1 (layer 2) then 2 (layer 0) fired DURING backward, in readiness order
        assert all_fired == [1, 2, 0]            # bucket 0 holds the unused layer 5: reduced by finish()
    assert np.array_equal(res[0][1][0], res[1][1][0]), "ranks must end with identical averaged gradients"
    from smilecode_amd.parallel import MODET_BUCKETS
    from smilecode_amd import synth
    for n in synth.param_spec():
        assert sum(any(n.startswith(q) for q in pre) for pre in MODET_BUCKETS) == 1, n


def _staged_worker(rank, world, port, q):
    import torch.distributed as dist
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank))
    from smilecode_amd.parallel import BucketedAllReduce, FlatParams, broadcast_parameters, init_from_env
    init_from_env("gloo")
    torch.manual_seed(7)
    # stand-in with ModeT's shape of dependencies: an "encoder" of two parts (buckets 2 and 1, the second fed by the first),
    # "heads" (bucket 0) that read BOTH encoder outputs -- cut with leaves exactly as engine.Trainer._staged_forward does
    net = torch.nn.ModuleDict({"heads": torch.nn.Linear(8, 3), "enc_hi": torch.nn.Linear(8, 4), "enc_lo": torch.nn.Linear(6, 8)})
    fp = FlatParams(net)
    broadcast_parameters(fp)
    bk = BucketedAllReduce(fp, list(net.named_parameters()), (("heads.",), ("enc_hi.",), ("enc_lo.",)))
    x = torch.full((5, 6), 0.1 * (rank + 1))

    def plain():
        fp.zero_grad()
        f1 = torch.tanh(net["enc_lo"](x)); f2 = torch.tanh(net["enc_hi"](f1))
        loss = net["heads"](torch.cat([f1[:, :4] * f2, f1[:, 4:]], 1)).square().sum()
        loss.backward()
        fp.gather_grads()
        return (fp.grad * fp.allreduce_grads()).clone()

    def staged():
        f1 = torch.tanh(net["enc_lo"](x))
        f1_hi = f1.detach().requires_grad_(True)                 # cut in front of the second encoder part
        f2 = torch.tanh(net["enc_hi"](f1_hi))
        l1, l2 = f1.detach().requires_grad_(True), f2.detach().requires_grad_(True)     # the heads see leaves
        loss = net["heads"](torch.cat([l1[:, :4] * l2, l1[:, 4:]], 1)).square().sum()
        fp.grad.fill_(float("nan"))
        bk.begin_staged()
        stages = [([loss], [None], [l1, l2]), None, None]
        carried = {}
        for k in range(3):
            params = [fp.params[i] for i in bk.members[k]]
            if k == 0:
                outs, gouts, extra = [loss], [None], [l1, l2]
            elif k == 1:
                outs, gouts, extra = [f2], [carried["l2"]], [f1_hi]
            else:
                outs, gouts, extra = [f1, f1], [carried["l1"], carried["f1_hi"]], []
            g = torch.autograd.grad(outs, params + extra, gouts, allow_unused=True)
            for i, gi in zip(bk.members[k], g[:len(params)]):
                off, n = fp.offsets[i]
                fp.grad[off:off + n].copy_(gi.reshape(-1))
            if k == 0:
                carried["l1"], carried["l2"] = g[len(params):]
            elif k == 1:
                carried["f1_hi"] = g[len(params)]
            bk.launch(k)                                         # all-reduce of bucket k while stage k + 1 runs
        return (fp.grad * bk.finish_staged()).clone()

    a, b = plain(), staged()
    q.put((rank, a.numpy(), b.numpy()))
    dist.destroy_process_group()


def test_staged_backward_with_bucket_allreduce_gloo_world2():
    """the graph-compatible form of cfg 5's overlapped all-reduce (engine.Trainer with overlap_allreduce=True): the backward
    as three autograd stages cut with leaves at the bucket boundaries, bucket k reduced right after stage k
    (BucketedAllReduce.begin_staged / launch / finish_staged) -- equals the plain backward + one all-reduce, on two gloo
    ranks with different data; every gradient slot is rewritten (NaN pre-fill)"""
    import torch.multiprocessing as mp
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 29411 + os.getpid() % 200
    procs = [ctx.Process(target=_staged_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = sorted([q.get(timeout=120) for _ in procs], key=lambda t: t[0])
    for p in procs:
        p.join(60)
        assert p.exitcode == 0
    for rank, plain, staged in res:
        assert np.isfinite(staged).all()
        assert np.allclose(plain, staged, rtol=0, atol=1e-6), np.abs(plain - staged).max()
    assert np.array_equal(res[0][2], res[1][2]), "ranks must end with identical averaged gradients"


def test_bench_roofline_traffic_sums_the_family_symbols():
    """roofline.traffic looked the FAMILY name up as a kernel symbol and quoted the minor variant's bytes (119 MB
    against 400 MB per launch of the real kernel, below the algorithmic bytes).  bench.family_traffic sums (bytes per launch x launches
    per step) over every symbol a family runs; add_traffic emits per-launch / per-step / ratio fields and flags a ratio below one."""
    sys.path.insert(0, ROOT)
    import bench
    pj = {"families": {"warp_bwd2_kernel": {"hbm_bytes_per_launch_corrected": 400e6, "launches_per_step": 4.0},
                       "warp_bwd_kernel": {"hbm_bytes_per_launch_corrected": 120e6, "launches_per_step": 3.0},
                       "fill_kernel": {"hbm_bytes_per_launch_corrected": 300e6, "launches_per_step": 4.0},
                       "accumulate_kernel": {"hbm_bytes_per_launch_corrected": 100e6, "launches_per_step": 6.0},
                       "adam_kernel": {"hbm_bytes_per_launch_corrected": 1e6, "launches_per_step": None}}}
    per_step, per_launch = bench.family_traffic(pj, "warp_bwd_kernel", 7.0)
    assert per_step == 4 * 400e6 + 3 * 120e6 and abs(per_launch - per_step / 7.0) < 1.0
    assert bench.family_traffic(pj, "warp_bwd_tiles", 6.0)[0] == 4 * 300e6 + 6 * 100e6           # (symbols absent from the profile add nothing)
    assert bench.family_traffic(pj, "no_such_family", 1.0) == (None, None)
    assert bench.family_traffic(pj, "adam_kernel", 1.0) == (None, None)                          # no per-step count in the profile
    r = {}
    bench.add_traffic(r, pj, "warp_bwd_kernel", 7.0, 1.1e9)
    assert r["traffic_per_step"] == 1.96e9 and abs(r["traffic_over_algorithmic"] - 1.96 / 1.1) < 1e-9 and "traffic_note" not in r
    r = {}
    bench.add_traffic(r, pj, "warp_bwd_kernel", 7.0, 5e9)
    assert r["traffic_over_algorithmic"] < 1.0 and "traffic_note" in r
    r = {}
    bench.add_traffic(r, None, "warp_bwd_kernel", 7.0, 1.1e9)
    assert r["traffic"] is None and r["traffic_per_step"] is None and r["algorithmic_bytes_per_step"] == 1.1e9
    assert bench.family("warp_bwd_tiles[C8]") == "warp_bwd_tiles" and bench.family("warp_bwd[C1]") == "warp_bwd_kernel"


def test_bench_self_launches_n_ranks_and_refuses_missing_gpus():
    """`python bench.py --gpus N` (no torchrun env) must itself start N ranks and prove it in the JSON line; with fewer
    than N GPUs it must fail loudly instead of silently running world=1.  Exercised on gloo with the
    all-reduce-only workload: same launcher, same rendezvous, no GPU needed."""
    import json
    env = dict(os.environ, MODET_DIST_BACKEND="gloo")
    env.pop("WORLD_SIZE", None)
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--workload", "allreduce", "--steps", "3",
                        "--warmup", "1"], env=env, capture_output=True, text=True, timeout=240)
    assert r.returncode == 0, r.stderr[-1500:]
    line = json.loads(r.stdout.strip().splitlines()[-1])
    assert line["n_gpus"] == 2 and line["world_size"] == 2 and line["backend"] == "gloo"
    assert line["allreduce"]["bytes"] == 1029670 * 4 and line["allreduce"]["us"] > 0
    if not torch.cuda.is_available():
        env.pop("MODET_DIST_BACKEND")
        r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "1"], env=env,
                           capture_output=True, text=True, timeout=120)
        assert r.returncode == 2 and "only 0 GPU(s)" in r.stderr and r.stdout.strip() == ""


def test_trainer_state_dict_is_torch_adam_compatible():
    """'optimizer' entry of the checkpoint (train.py:158-163): what we save loads into a real torch.optim.Adam(amsgrad) and
    what torch saves loads into the Trainer (host-side logic, runs on CPU tensors)"""
    from smilecode_amd.engine import Trainer
    torch.manual_seed(0)
    net = torch.nn.Sequential(torch.nn.Linear(5, 7), torch.nn.Linear(7, 3))
    tr = Trainer(net, lr=1e-4)
    assert tr.state_dict()["state"] == {}                                 # nothing stepped yet, like a fresh Adam
    tr.m.normal_(); tr.v.uniform_(0.1, 1.0); tr.vmax.copy_(tr.v * 2)
    tr.step, tr.lr_last = 3, 5e-5
    sd = tr.state_dict()
    opt = torch.optim.Adam(net.parameters(), lr=1e-4, weight_decay=0, amsgrad=True)
    opt.load_state_dict(sd)                                               # strict: raises on any format mismatch
    for i, p in enumerate(net.parameters()):
        off, k = tr.fp.offsets[i]
        st = opt.state[p]
        assert float(st["step"]) == 3.0
        assert torch.equal(st["exp_avg"].reshape(-1), tr.m[off:off + k])
        assert torch.equal(st["max_exp_avg_sq"].reshape(-1), tr.vmax[off:off + k])
    assert opt.param_groups[0]["lr"] == 5e-5 and opt.param_groups[0]["amsgrad"] is True
    # a torch optimizer that really stepped -> Trainer
    for p in net.parameters():
        p.grad = torch.randn_like(p)
    opt.step()
    tr2 = Trainer(torch.nn.Sequential(torch.nn.Linear(5, 7), torch.nn.Linear(7, 3)))
    tr2.load_state_dict(opt.state_dict())
    assert tr2.step == 4
    sd2 = tr2.state_dict()
    for i, p in enumerate(net.parameters()):
        for key in ("exp_avg", "exp_avg_sq", "max_exp_avg_sq"):
            assert torch.equal(sd2["state"][i][key], opt.state[p][key])
    with pytest.raises(RuntimeError):
        tr2.load_state_dict({"step": 1, "lr": 1e-4})                       # round 1's stub format is rejected loudly


def test_trainer_picks_the_seeded_backward_only_for_the_losses_its_kernels_cover():
    """host logic of engine.Trainer._seedable: the step seeds its backward with the loss kernels' gradients only for the
    reference's two terms as the kernels have them -- cubic NCC window 3 / 5 / 7 / 9, Grad3d without loss_mult -- on a model that
    hands out channels-last results; everything else keeps the autograd expression (no silent approximation)"""
    from smilecode_amd import engine, losses, models

    class WithCl(torch.nn.Linear):
        def forward_cl(self, a, b):
            raise AssertionError("not called here")

    tr = engine.Trainer(WithCl(3, 2))
    assert tr._seedable()
    tr.seed_backward = False
    assert not tr._seedable()
    tr.seed_backward = True
    for sim, ok in ((losses.NCC_vxm(win=[7, 7, 7]), True), (losses.NCC_vxm(win=[5, 3, 7]), False),
                    (losses.NCC_vxm(win=[11, 11, 11]), False), (losses.NCC_vxm(win=[4, 4, 4]), False)):
        tr.sim = sim
        assert tr._seedable() == ok, sim._w
    tr.sim = losses.NCC_vxm()
    tr.reg = losses.Grad3d(penalty="l2", loss_mult=2)
    assert not tr._seedable()
    tr.reg = losses.Grad3d(penalty="l1")
    assert tr._seedable()

    class Sub(losses.Grad3d):                            # a subclass may compute anything: not the kernel's contract
        pass
    tr.reg = Sub()
    assert not tr._seedable()
    assert not engine.Trainer(torch.nn.Linear(3, 2))._seedable()          # a model without forward_cl
    assert hasattr(models.ModeT, "forward_cl") and hasattr(models.ModeT_cu, "forward_cl")


def test_resume_from_a_checkpoint_the_reference_wrote(tmp_path):
    """the reference's train.py stores ``param_groups[0]['lr'] = round(INIT_LR * np.power(..), 8)`` -- a
    numpy.float64 (train.py:117,:166-168) -- in the optimizer state it saves (train.py:158-163).  torch >= 2.6 refuses to
    unpickle that with its default weights_only=True; smilecode_amd.train / infer load with weights_only=False and the
    Trainer casts the scalars."""
    import inspect
    from smilecode_amd import infer, train
    from smilecode_amd.engine import Trainer
    torch.manual_seed(1)
    net = torch.nn.Sequential(torch.nn.Linear(5, 7), torch.nn.Linear(7, 3))
    opt = torch.optim.Adam(net.parameters(), lr=1e-4, weight_decay=0, amsgrad=True)
    for p in net.parameters():
        p.grad = torch.randn_like(p)
    opt.step()
    lr = round(1e-4 * np.power(1 - 3 / 30, 0.9), 8)                       # adjust_learning_rate, train.py:166-168
    assert type(lr).__module__ == "numpy"
    for g in opt.param_groups:
        g["lr"] = lr
    path = str(tmp_path / "dsc0.612.pth.tar")
    torch.save({"epoch": 4, "state_dict": net.state_dict(), "best_dsc": np.float64(0.612), "optimizer": opt.state_dict()}, path)
    with pytest.raises(Exception):
        torch.load(path, map_location="cpu")                              # the failure the advisor reproduced
    ck = torch.load(path, map_location="cpu", weights_only=False)
    tr = Trainer(torch.nn.Sequential(torch.nn.Linear(5, 7), torch.nn.Linear(7, 3)))
    tr.load_state_dict(ck["optimizer"])
    assert tr.step == 1 and type(tr.lr_last) is float and abs(tr.lr_last - float(lr)) < 1e-15
    assert type(tr.state_dict()["param_groups"][0]["lr"]) is float        # what we save loads under weights_only=True
    for mod in (train, infer):
        src = inspect.getsource(mod.main)
        assert "torch.load(" in src and src.count("weights_only=False") >= src.count("torch.load(")


def test_flat_params_gather_with_directly_written_gradients():
    """FlatParams.gather_grads(written): parameters whose gradient was written straight into the flat buffer (the
    deferred weight-gradient reductions) are left alone, a further autograd contribution to such a parameter is added
    on top, the others are copied, and parameters without any gradient are zeroed."""
    import torch.nn as nn
    from smilecode_amd.parallel import FlatParams

    class M(nn.Module):
        def __init__(self):
            super().__init__()
            self.a = nn.Parameter(torch.arange(6.0).reshape(2, 3))
            self.b = nn.Parameter(torch.ones(4))
            self.c = nn.Parameter(torch.full((5,), 2.0))
            self.d = nn.Parameter(torch.zeros(3))

    m = M()
    fp = FlatParams(m)
    dst = fp.grad_destinations()
    assert set(dst) == {p.data_ptr() for p in fp.params} and dst[m.a.data_ptr()].shape == m.a.shape
    assert dst[m.b.data_ptr()].data_ptr() == fp.grad[6:10].data_ptr()          # views of the flat buffer, in order
    fp.grad.fill_(-7.0)                                                        # stale values from the previous step
    fp.zero_grad()
    dst[m.a.data_ptr()].copy_(torch.full((2, 3), 10.0))                        # written directly
    dst[m.b.data_ptr()].copy_(torch.full((4,), 20.0))                          # written directly ...
    m.b.grad = torch.full((4,), 1.0)                                           # ... plus a second use through autograd
    m.c.grad = torch.full((5,), 3.0)                                           # ordinary autograd gradient
    fp.gather_grads({m.a.data_ptr(), m.b.data_ptr()})
    assert torch.equal(fp.grad, torch.cat([torch.full((6,), 10.0), torch.full((4,), 21.0), torch.full((5,), 3.0),
                                           torch.zeros(3)]))
    fp.zero_grad()
    m.a.grad = torch.ones(2, 3)
    fp.gather_grads()                                                          # no direct writes: everything as before
    assert torch.equal(fp.grad[:6], torch.ones(6)) and float(fp.grad[6:].abs().sum()) == 0.0


def test_save_checkpoint_rotation_keeps_the_eight_best(tmp_path):
    """save_checkpoint (train.py:171-176): at most 8 files, the naturally-sorted first (lowest Dice) goes first"""
    from smilecode_amd.train import latest_checkpoint, save_checkpoint
    d = str(tmp_path) + "/"
    dscs = [0.512, 0.498, 0.530, 0.100, 0.527, 0.610, 0.605, 0.590, 0.611, 0.045, 0.700]
    for v in dscs:
        save_checkpoint({"epoch": 1, "dsc": v}, save_dir=d, filename="dsc{:.3f}.pth.tar".format(v))
    left = sorted(os.listdir(d))
    assert len(left) == 8
    assert left == sorted("dsc{:.3f}.pth.tar".format(v) for v in sorted(dscs)[-8:])
    assert latest_checkpoint(d).endswith("dsc0.700.pth.tar")             # what --cont-training resumes from (train.py:83)


def test_device_volume_cache_indexes_pairs_like_the_datasets(tmp_path):
    """DeviceVolumeCache (here on the CPU device) hands out the same pairs as the reference-shaped datasets"""
    import pickle
    from smilecode_amd import data
    rng = np.random.default_rng(0)
    for i in range(4):
        with open(tmp_path / f"s{i}.pkl", "wb") as f:
            pickle.dump((rng.random((4, 6, 5), dtype=np.float32), rng.choice([0, 21, 34, 166, 999], (4, 6, 5)).astype(np.uint16)), f)
    ds = data.LPBABrainInferDatasetS2S([str(p) for p in tmp_path.glob("*.pkl")])
    cache = data.DeviceVolumeCache(ds, device="cpu", with_labels=True, workers=2)
    assert len(cache) == len(ds) == 12
    for i in range(len(ds)):
        want, got = ds[i], cache.pair(i)
        for w, g in zip(want, got):
            assert g.shape == (1,) + tuple(w.shape) and g.dtype == w.dtype and torch.equal(g[0], w)
    syn = data.SyntheticPairs((8, 8, 8), 3, 24)
    c2 = data.DeviceVolumeCache(syn, device="cpu")
    x, y = c2.pair(4)
    assert torch.equal(x[0], syn[4][0]) and torch.equal(y[0], syn[4][1])


def test_data_pipeline_golden_from_the_reference(tmp_path):
    """f2 pinned: pair order, volumes and Seg_norm labels equal what the reference's data/datasets.py + data/trans.py returned
    (tests/golden/data_pipeline.npz, make_goldens_data.py); the cache on the CPU device here, on the GPU in test_gpu_e2e.py"""
    from tests.util import check_data_pipeline_golden
    assert check_data_pipeline_golden(tmp_path, "cpu") == 4


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


def test_oracle_eval_goldens(orc):
    """oracle restatements of jacobian_determinant_vxm / Grad3d('l1') / first-argument NCC against the vectors captured
    from the reference itself (tests/golden/make_goldens_eval.py)"""
    g = gold("op_eval.npz")
    for tag in "abc":
        det = orc.jacobian_determinant(g[f"jac.{tag}.flow"])
        assert np.array_equal(det, g[f"jac.{tag}.det"]), "same fp64 operations in the same order: bit-identical"
        assert int(np.sum(det <= 0)) == int(g[f"jac.{tag}.nonpos"][0])
    disp = np.zeros((3, 6, 7, 8), np.float32)
    assert np.array_equal(orc.jacobian_determinant(disp), np.ones((6, 7, 8)))
    disp[0] = -2.0 * np.arange(6)[:, None, None]            # x -> -x: folding
    assert (orc.jacobian_determinant(disp) < 0).all()
    f = T(g["g3d_l1.flow"]).requires_grad_(True)
    l = orc.grad3d_loss(f, "l1")
    assert abs(float(l) - float(g["g3d_l1.val"])) < 1e-12
    assert_close(torch.autograd.grad(l, f)[0].numpy(), g["g3d_l1.dflow"], atol=1e-15, rtol=1e-12, what="grad3d l1 dflow")
    a = T(g["ncc1.a"]).requires_grad_(True)
    l = orc.ncc_loss(a, T(g["ncc1.b"]))
    assert abs(float(l) - float(g["ncc1.val"])) < 1e-12
    assert_close(torch.autograd.grad(l, a)[0].numpy(), g["ncc1.da"], atol=1e-15, rtol=1e-9, what="ncc d y_true")
    for w in (3, 5, 7):                                     # NCC_vxm(win=[w, w, w]), from the reference's own class
        a, b = T(g[f"nccw{w}.a"]).requires_grad_(True), T(g[f"nccw{w}.b"]).requires_grad_(True)
        l = orc.ncc_loss(a, b, win=w)
        assert abs(float(l) - float(g[f"nccw{w}.val"])) < 1e-12
        da, db = torch.autograd.grad(l, [a, b])
        assert_close(da.numpy(), g[f"nccw{w}.da"], atol=1e-15, rtol=1e-9, what=f"ncc win {w} d y_true")
        assert_close(db.numpy(), g[f"nccw{w}.db"], atol=1e-15, rtol=1e-9, what=f"ncc win {w} d y_pred")


def test_oracle_ncc_any_window_goldens(orc):
    """the oracle's NCC for even / anisotropic / > 9-voxel windows (every axis padded by floor(win[0] / 2), losses.py:57) against
    the vectors captured from the reference's own class (tests/golden/make_goldens_ncc_windows.py)"""
    g = gold("op_ncc_windows.npz")
    for w in ([4, 4, 4], [5, 3, 7], [11, 11, 11], [2, 6, 3], [6, 9, 9], [9, 9, 5], [1, 1, 1]):
        tag = "x".join(map(str, w))
        a, b = T(g[f"ncc[{tag}].a"]).requires_grad_(True), T(g[f"ncc[{tag}].b"]).requires_grad_(True)
        l = orc.ncc_loss(a, b, win=w)
        assert abs(float(l) - float(g[f"ncc[{tag}].val"])) < 1e-12
        da, db = torch.autograd.grad(l, [a, b])
        assert_close(da.numpy(), g[f"ncc[{tag}].da"], atol=1e-15, rtol=1e-9, what=f"ncc win {w} d y_true")
        assert_close(db.numpy(), g[f"ncc[{tag}].db"], atol=1e-15, rtol=1e-9, what=f"ncc win {w} d y_pred")


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


def test_gradient_maximum_tag_dies_with_in_place_accumulation():
    """ops._tag_amax / _amax_of (round 5): the InstanceNorm backward tags its output with max |d_x| so that the conv backward that
    consumes it may split it into f16 pieces; the tag is valid only while the tensor is what the kernel wrote -- autograd's
    in-place gradient accumulation bumps the version counter and must kill it (a stale maximum would overflow f16)."""
    from torch.autograd import Function

    from smilecode_amd import ops
    seen = []

    class Consumer(Function):
        @staticmethod
        def forward(ctx, x):
            return x * 2

        @staticmethod
        def backward(ctx, g):
            seen.append(ops._amax_of(g))
            return g * 2

    class Producer(Function):
        @staticmethod
        def forward(ctx, x):
            return x + 1

        @staticmethod
        def backward(ctx, g):
            out = g.clone()
            return ops._tag_amax(out, out.abs().max().reshape(1))

    x = torch.randn(5, requires_grad=True)
    Producer.apply(Consumer.apply(x)).sum().backward()
    a = Consumer.apply(x)
    (Producer.apply(a) + Producer.apply(a)).sum().backward()       # two gradients accumulated into one tensor
    assert seen[0] is not None and float(seen[0]) == 1.0
    assert seen[1] is None
    t = ops._tag_amax(torch.zeros(3), torch.ones(1))
    assert ops._amax_of(t) is not None
    t.add_(1)
    assert ops._amax_of(t) is None
