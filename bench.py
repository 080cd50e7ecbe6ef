#!/usr/bin/env python
"""Headline benchmark: ModeT volume-pairs/sec on synthetic 160x192x160 fp32 pairs (BASELINE.json).

    python bench.py [--gpus N] [--steps K] [--warmup W] [--workload train|fwd|op|allreduce]
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...

A "step" = one pass of the hot path over one volume pair per rank:
  train (default, BASELINE metric "fwd+bwd"): ModeT.forward + NCC + Grad3d('l2') + backward +
        (RCCL all-reduce of the flat 4.12 MB gradient when N>1) + Adam-amsgrad step   [cfg 3 / cfg 4]
  fwd : ModeT.forward incl. the final warp, no grad                                    [cfg 2]
Inputs are resident in HBM before the timed region.  One JSON line on stdout (rank 0).

roofline: the dominant kernel group (largest share of step time, found in an untimed profiled step) is
timed live with HIP events on the launch stream during the K timed steps; achieved = its algorithmic
FLOPs (or bytes) / its summed duration.  cpu_baseline: the CPU oracle (oracle/modet_torch.py, ATen-CPU,
same op sequence as the reference's PyTorch-CPU path) on this host's cores, rank 0, N=1 only.
"""
from __future__ import annotations

import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

import torch  # noqa: E402
import torch.distributed as dist  # noqa: E402

PEAK_MFMA_F32_TFLOPS = 157.3      # MI355X_MICROARCH.md: v_mfma_f32_16x16x4_f32, dense
PEAK_HBM_GBS = 8000.0             # MI355X_MICROARCH.md: HBM3E spec


def log(*a):
    print(*a, file=sys.stderr, flush=True)


def family(tag):
    """launch tag -> kernel family (= the kernel symbol rocprof reports, template arguments stripped).  The conv tags carry
    the family the library picked for that shape (ops._conv_tag): @x3 = z-marching bf16x3, @split = tiled bf16x3, @direct = the small-volume direct MFMA kernel, @tr = the transpose-read bf16x3 weight gradient."""
    base = tag.split("[")[0]
    if tag.startswith("conv_fwd[1->"):
        return "conv_c1_fwd_kernel"
    if tag.startswith("conv_wgrad[1->"):
        return "conv_c1_wgrad_mfma_kernel"
    if base in ("conv_bf16_fwd", "conv_bf16_dgrad"):        # one bf16 piece: HBM-bound, priced against the HBM roofline
        return "conv_x3_kernel<NPC=1>" if tag.endswith("@x3") else "conv3d_bf16_kernel"
    if base == "conv_bf16_wgrad":
        return "conv_x3_wgrad_kernel<NPC=1>" if tag.endswith("@x3") else "conv3d_bf16_wgrad_kernel"
    if base in ("conv_fwd", "conv_dgrad"):
        if tag.endswith("@direct"):
            return "conv_direct_kernel"
        if tag.endswith(("@q", "@qh")):
            return "conv_q_kernel"
        return "conv_x3_kernel" if tag.endswith(("@x3", "@x3h")) else ("conv3d_bf16_kernel<SP=3>" if tag.endswith("@split") else "conv3d_mfma_kernel")
    if base == "conv_wgrad":
        return "conv_x3_wgrad_kernel" if tag.endswith(("@x3", "@x3h")) else ("conv_wgrad_tr_kernel" if tag.endswith(("@tr", "@trh")) else "conv3d_wgrad_kernel")
    if base == "warp_bwd":
        return "warp_bwd_kernel"
    if base == "warp_bwd_tiles":
        return "warp_bwd_tiles"
    if base == "warp_bwd_gather3":
        return "warp_bwd_gather3_tiled_kernel"
    return base


# kernel family -> the rocprof kernel SYMBOLS (template arguments stripped) its launches run: a family is what one launch tag of
# ops.py enqueues, which may be several kernels (the destination-tile warp backward is six) or one of several variants of a
# kernel (warp_bwd_kernel / warp_bwd2_kernel).  roofline.traffic sums the counter bytes of ALL of them (the
# family name looked up as a symbol found the minor variant only).  Families not listed are their own symbol.
FAMILY_SYMBOLS = {
    "warp_bwd_kernel": ("warp_bwd_kernel", "warp_bwd2_kernel"),
    "warp_bwd_tiles": ("fill_kernel", "fill_c3_kernel", "accumulate_kernel", "border_kernel"),
    "warp_fwd": ("warp_fwd_kernel",),
    "instnorm_lrelu_bwd": ("in_bwd_apply_kernel", "in_partial_kernel", "in_rows_finalize_bwd_kernel"),
    "instnorm_lrelu_fwd": ("in_apply_kernel", "in_apply_pool_kernel", "in_rows_finalize_kernel", "in_finalize_kernel"),
    "proj_ln_bwd": ("proj_ln_bwd_g_kernel", "proj_ln_bwd_kernel"),
    "proj_ln_fwd": ("proj_ln_fwd_g_kernel", "proj_ln_fwd_kernel"),
    "na_bwd": ("na_bwd_march_kernel", "na_bwd_kernel"),
    "na_fwd": ("na_fwd_kernel",),
    "ncc_fwd_bwd": ("ncc_march_kernel",),
    "grad3d_fwd_bwd": ("grad3d_kernel",),
    "upsample2_bwd": ("upsample2_bwd_kernel", "upsample2_bwd_rows_kernel", "upsample2_bwd_axis_kernel"),
    "upsample2_fwd": ("upsample2_fwd_kernel",),
    "conv_c1_fwd_kernel": ("conv_c1_march_kernel",),
    "conv_x3_kernel<NPC=1>": ("conv_x3_kernel",),
    "conv_x3_wgrad_kernel<NPC=1>": ("conv_x3_wgrad_kernel",),
}


def family_traffic(pj, fam, launches_per_step):
    """HBM bytes of one family from profiles/pmc_traffic.json: (bytes per step, bytes per launch of the family) or (None, None).
    Per step = sum over the family's kernel symbols of (mean bytes per launch of the symbol) x (its launches per step)"""
    fams = pj.get("families", {})
    tot, found = 0.0, False
    for sym in FAMILY_SYMBOLS.get(fam, (fam,)):
        v = fams.get(sym)
        if v and v.get("launches_per_step"):
            tot += v["hbm_bytes_per_launch_corrected"] * v["launches_per_step"]
            found = True
    if not found:
        return None, None
    return tot, (tot / launches_per_step if launches_per_step else None)


def add_traffic(r, pj, fam, launches_per_step, alg_bytes_per_step):
    """traffic fields of a roofline object (r['traffic'] stays per launch, as 'achieved' is)"""
    r["algorithmic_bytes_per_step"] = alg_bytes_per_step
    per_step, per_launch = (None, None) if pj is None else family_traffic(pj, fam, launches_per_step)
    r["traffic"], r["traffic_per_step"] = per_launch, per_step
    r["traffic_over_algorithmic"] = (per_step / alg_bytes_per_step) if (per_step and alg_bytes_per_step) else None
    if r["traffic_over_algorithmic"] is not None and r["traffic_over_algorithmic"] < 1.0:
        # possible physically (the 256 MB Infinity Cache serves a tensor the previous kernel just wrote) -- but say so
        r["traffic_note"] = ("counter traffic below the algorithmic bytes: part of this family's input was served by the 256 MB "
                             "Infinity Cache (written by the kernel before it) or a symbol of the family is missing from the profile")
        log(f"[bench] note: {fam}: counter traffic {per_step / 1e6:.1f} MB/step < algorithmic {alg_bytes_per_step / 1e6:.1f} MB/step")


# fp32-accurate matrix work on the 16-bit pipe costs six bf16 MFMAs per fp32 product (three bf16 pieces per operand,
# csrc/conv3d_x3.hip), or three f16 MFMAs (two f16 pieces per operand: the FORWARD launches of the z-marching kernel since round
# 5; f16 and bf16 MFMA have the same dense peak): the ceiling of a family in algorithmic (fp32) FLOP/s is the dense 16-bit peak
# divided by the FLOP-weighted mean number of piece products of its launches
PEAK_MFMA_BF16_TFLOPS = 2500.0


def piece_products(tag):
    """MFMA piece products per fp32 product of one launch tag"""
    return 3.0 if tag.endswith(("@x3h", "@qh", "@trh")) else 6.0         # ops._conv_tag: 'h' = the launch ran on two f16 pieces
MFMA_F32_FAMILIES = ("conv3d_mfma_kernel", "conv3d_wgrad_kernel", "conv_c1_wgrad_mfma_kernel", "conv_direct_kernel")
MFMA_X3_FAMILIES = ("conv_x3_kernel", "conv_x3_wgrad_kernel", "conv_wgrad_tr_kernel", "conv_q_kernel", "conv3d_bf16_kernel<SP=3>")


def csrc_sha16():
    """fingerprint of the kernel sources (same function as tools/pmc_traffic.py)"""
    import hashlib
    root = os.path.join(ROOT, "smilecode_amd", "csrc")
    h = hashlib.sha256()
    for fn in sorted(os.listdir(root)):
        if fn.endswith((".hip", ".h")):
            h.update(fn.encode())
            h.update(open(os.path.join(root, fn), "rb").read())
    return h.hexdigest()[:16]


def roof_of(fam, flops, nbytes, sec, pflops=None):
    """roofline object of one kernel family from its algorithmic work and summed duration (pflops: the launches' FLOPs times
    their piece products, summed -- the MFMA work actually issued).  A matrix kernel is priced against BOTH roofs -- the time
    its MFMA work needs at the pipe's peak and the time its algorithmic bytes need at the HBM peak -- and the larger one is its
    bound: since the level-1 convolutions run on two f16 pieces (round 5) their 64 bytes per voxel outweigh their matrix work."""
    hbm = {"bound": "hbm", "achieved": nbytes / sec / 1e9, "peak": PEAK_HBM_GBS, "unit": "GB/s", "frac": nbytes / sec / 1e9 / PEAK_HBM_GBS}
    if fam in MFMA_F32_FAMILIES:
        ach = flops / sec / 1e12
        r = {"bound": "mfma", "achieved": ach, "peak": PEAK_MFMA_F32_TFLOPS, "unit": "TFLOP/s", "frac": ach / PEAK_MFMA_F32_TFLOPS}
    elif fam in MFMA_X3_FAMILIES:
        prod = (pflops / flops) if (pflops and flops) else 6.0
        ach, peak = flops / sec / 1e12, PEAK_MFMA_BF16_TFLOPS / prod
        r = {"bound": "mfma", "achieved": ach, "peak": peak, "unit": "TFLOP/s", "frac": ach / peak,
             "piece_products_per_fp32_product": prod,
             "peak_note": "fp32-accurate FLOPs on the 16-bit pipe: dense bf16 / f16 MFMA peak 2500 TFLOP/s / the FLOP-weighted "
                          "piece products per fp32 product of these launches (6 = three bf16 pieces per operand, 3 = two f16 "
                          "pieces); against the exact-f32 MFMA peak (157.3) the same number is frac_of_f32_mfma_peak",
             "frac_of_f32_mfma_peak": ach / PEAK_MFMA_F32_TFLOPS}
    else:
        return hbm
    if hbm["frac"] > r["frac"]:                   # the HBM roof is the tighter one for this family's launches
        hbm["other_roof"] = {k: r[k] for k in ("bound", "achieved", "peak", "unit", "frac")}
        return hbm
    r["other_roof"] = {k: hbm[k] for k in ("bound", "achieved", "peak", "unit", "frac")}
    return r


def cpu_baseline(shape, workload, budget_s=40.0):
    """oracle timed on the host cores on a bounded sample of the same workload"""
    from oracle import modet_torch as orc
    from smilecode_amd import synth
    heads = (8, 4, 2, 1, 1)
    cores = torch.get_num_threads()

    def run(shp):
        p = {n: torch.from_numpy(v).requires_grad_(workload == "train") for n, v in synth.make_weights(24).items()}
        mov, fix = (torch.from_numpy(a) for a in synth.make_pair(shp, 24))
        t0 = time.perf_counter()
        if workload == "train":
            loss = orc.train_loss(p, mov, fix, heads, 6, 1.0)[0]
            torch.autograd.grad(loss, list(p.values()))
        else:
            with torch.no_grad():
                orc.modet_forward(p, mov, fix, heads, 6, 1.0)
        return time.perf_counter() - t0

    half = tuple(s // 2 for s in shape)
    run((32, 48, 32))                       # warm ATen / thread pool
    # ATen-CPU does not scale to every core of a big host on these shapes: time the bounded sample at two thread
    # counts and report the faster one (cores = threads actually used)
    best = None
    for nt in sorted({cores, min(cores, 32)}):
        torch.set_num_threads(nt)
        t = run(half)
        if best is None or t < best[0]:
            best = (t, nt)
    t_half, nt = best
    torch.set_num_threads(nt)
    if 8.0 * t_half * 1.2 <= budget_s:
        t_full = run(shape)
        out = {"value": 1.0 / t_full, "unit": "volume-pairs/sec", "cores": nt, "kind": "port",
               "sample": "1 pair %dx%dx%d %s, oracle/modet_torch.py (ATen-CPU fp32), %.2f s" % (*shape, workload, t_full)}
    else:
        out = {"value": 1.0 / (8.0 * t_half), "unit": "volume-pairs/sec", "cores": nt, "kind": "port",
               "sample": "1 pair %dx%dx%d %s (1/8 of the voxels, %.2f s) scaled x8, oracle/modet_torch.py (ATen-CPU fp32)"
                         % (*half, workload, t_half)}
    # the one-thread leg BASELINE.md section 4(2) asks for: the same oracle on ONE core, on the largest sample of the workload
    # that one core finishes in about half a minute (the full pair would take minutes), scaled by voxels
    try:
        torch.set_num_threads(1)
        tiny = (32, 48, 32)
        t_tiny = run(tiny)
        vox = lambda s: s[0] * s[1] * s[2]                               # noqa: E731
        sample = half if t_tiny * vox(half) / vox(tiny) <= 45.0 else (64, 64, 64)
        t1 = run(sample)
        scale = vox(shape) / vox(sample)
        out["value_1_thread"] = 1.0 / (t1 * scale)
        out["sample_1_thread"] = ("1 pair %dx%dx%d %s on one thread (%.2f s) scaled x%.2f by voxels, oracle/modet_torch.py (ATen-CPU fp32)"
                                  % (*sample, workload, t1, scale))
    except Exception as e:                                                # noqa: BLE001
        out["value_1_thread"], out["sample_1_thread"] = None, f"failed: {e!r}"
    torch.set_num_threads(cores)
    return out


def self_launch(n):
    """`python bench.py --gpus N` without a torchrun environment: re-exec this script under torch.distributed.run with one
    rank per GPU (what the driver's multi-GPU tier does itself).  Fails loudly when the node has fewer than N devices."""
    import socket
    import subprocess
    backend = os.environ.get("MODET_DIST_BACKEND") or "nccl"
    have = torch.cuda.device_count() if torch.cuda.is_available() else 0
    if backend == "nccl" and have < n:
        log(f"[bench] ERROR: --gpus {n} requested but only {have} GPU(s) are visible on this node; "
            "one rank per GPU is required (RCCL over xGMI), refusing to oversubscribe")
        sys.exit(2)
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={n}", "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    log(f"[bench] --gpus {n} without WORLD_SIZE: launching {n} ranks: {' '.join(cmd)}")
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0")
    sys.exit(subprocess.run(cmd, env=env).returncode)


def allreduce_probe(buf, world, iters=20):
    """average time of one all-reduce(SUM) of the flat gradient buffer and the bus bandwidth it implies
    (ring convention: 2*(world-1)/world * bytes / t); HIP events on the current stream when the buffer is on a GPU"""
    if world < 2:
        return None
    for _ in range(3):
        dist.all_reduce(buf)
    if buf.is_cuda:
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        dist.barrier()
        e0.record()
        for _ in range(iters):
            dist.all_reduce(buf)
        e1.record()
        e1.synchronize()
        us = e0.elapsed_time(e1) * 1e3 / iters
    else:
        dist.barrier()
        t0 = time.perf_counter()
        for _ in range(iters):
            dist.all_reduce(buf)
        us = (time.perf_counter() - t0) * 1e6 / iters
    t = torch.tensor([us], dtype=torch.float64, device=buf.device)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    us = float(t.item())
    nbytes = buf.numel() * buf.element_size()
    return {"bytes": nbytes, "us": us, "bus_GBps": 2.0 * (world - 1) / world * nbytes / (us * 1e-6) / 1e9,
            "algbw_GBps": nbytes / (us * 1e-6) / 1e9}


def allreduce_only(args, rank, local, world):
    """--workload allreduce: only the step's one exchange -- the flat 1 029 670-float (4.12 MB) gradient all-reduce --
    timed K times.  Runs on any backend (gloo on CPU in the tests), so the N-rank launch path can be exercised
    without GPUs."""
    on_gpu = dist.is_initialized() and dist.get_backend() == "nccl"
    dev = torch.device("cuda", local) if on_gpu else torch.device("cpu")
    buf = torch.ones(1029670, dtype=torch.float32, device=dev)
    probe = allreduce_probe(buf, world, iters=max(args.steps, 1))
    if world > 1:
        assert float(buf[0]) > 1.0                       # the ranks really summed
    if rank == 0:
        print(json.dumps({"metric": "gradient all-reduce (1 029 670 fp32)", "value": probe["us"] if probe else None,
                          "unit": "us", "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "higher_is_better": False,
                          "backend": dist.get_backend() if dist.is_initialized() else None,
                          "world_size": dist.get_world_size() if dist.is_initialized() else 1,
                          "allreduce": probe, "data": "synthetic", "dtype": "f32",
                          "config": {"workload": "flat gradient all-reduce only", "parallelism": f"dp{world}"}}), flush=True)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


def operator_only(args, rank, local, world):
    """--workload op: the reference's own operator boundary on its own -- ``modetqkrpb_cu`` forward + backward
    (smilecode_amd/functional.py -> modet_qk_fwd / modet_qk_bwd) at the five shapes one ModeT-cu training step calls it
    with (ModeT-cu/models.py:323-352).  One step = the five forward and five backward calls; value = steps/s = volume
    pairs/s through the operator alone.  roofline = the level-1 backward (the longest launch), algorithmic bytes
    d_attn 27 + q 6 + kpad 6 + d_q 6 + d_kpad 6 elements per voxel, HIP events on the launch stream."""
    import importlib.util
    spec = importlib.util.spec_from_file_location("bench_operator", os.path.join(os.path.dirname(os.path.abspath(__file__)),
                                                                                "tools", "bench_operator.py"))
    bo = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(bo)
    torch.cuda.set_device(local)
    rows = bo.run(max(args.steps, 3))
    ms = sum(r["fwd_ms"] + r["bwd_ms"] for r in rows)
    t = torch.tensor([ms], dtype=torch.float64, device="cuda")
    if world > 1:
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
    ms = float(t)
    l1 = rows[0]
    cpu = None
    if rank == 0 and not args.no_cpu_baseline:
        # the plain-C restatement of the CUDA operator (oracle/modet_ref.c: ref_modet_fw / ref_modet_bw, fp64, one thread)
        # on a bounded sample -- the level-1 shape, 4 915 200 of the step's 5 768 400 voxel-heads -- scaled to the whole step
        import numpy as np
        from oracle import cref
        rng = np.random.default_rng(0)
        D, H, W = 160, 192, 160
        q = rng.standard_normal((1, 1, D, H, W, 6))
        kp = np.zeros((1, 1, D + 2, H + 2, W + 2, 6))
        kp[:, :, 1:-1, 1:-1, 1:-1] = rng.standard_normal((1, 1, D, H, W, 6))
        rpb = rng.standard_normal((1, 3, 3, 3))
        ga = rng.standard_normal((1, 1, D, H, W, 27))
        t0 = time.perf_counter()
        cref.modet_fw(q, kp, rpb)
        cref.modet_bw(ga, q, kp, True)
        dt = time.perf_counter() - t0
        vh_step = sum(r["shape"][0] * r["shape"][1] * r["shape"][2] * r["heads"] for r in rows)
        cpu = {"value": 1.0 / (dt * vh_step / (D * H * W)), "unit": "volume-pairs/sec", "cores": 1, "kind": "port",
               "sample": "modet_fw + modet_bw of oracle/modet_ref.c (plain C, fp64) at 160x192x160, 1 head: %.2f s, scaled by voxel-heads"
                         % dt}
    traffic = None
    try:                                             # per-launch HBM bytes of the same kernel from the rocprofv3 --pmc passes
        with open(os.path.join(ROOT, "profiles", "r02f_pmc_traffic_operator.json")) as f:
            traffic = json.load(f)["families"]["qk_bwd_plane_kernel"]["hbm_bytes_per_launch_corrected"]
    except (OSError, KeyError, ValueError):
        pass
    if rank == 0:
        print(json.dumps({
            "metric": "volume-pairs/sec through modetqkrpb_cu fwd+bwd (5 levels of 160x192x160)", "value": world * 1e3 / ms,
            "unit": "volume-pairs/sec", "n_gpus": world, "steps": args.steps, "warmup": 3, "ms_per_step": ms,
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "f32", "data": "synthetic",
            "config": {"workload": "modetqkrpb_cu operator only, head_dim 6, heads 8/4/2/1/1, batch 1/GPU, median of per-call HIP events"},
            "roofline": {"bound": "hbm", "kernel": "qk_bwd_plane_kernel<float,6> @160x192x160", "achieved": l1["bwd_GBps"],
                         "peak": 8000.0, "unit": "GB/s", "frac": l1["bwd_GBps"] / 8000.0, "traffic": traffic},
            "cpu_baseline": cpu, "levels": rows}), flush=True)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


def cpu_cfg1():
    """BASELINE.json configs[0]: single 64x64x64 synthetic pair, CPU forward of the oracle (ATen-CPU, the reference's op
    sequence), at all host threads and at one thread"""
    from oracle import modet_torch as orc
    from smilecode_amd import synth
    p = {n: torch.from_numpy(v) for n, v in synth.make_weights(24).items()}
    mov, fix = (torch.from_numpy(a) for a in synth.make_pair((64, 64, 64), 24))
    cores = torch.get_num_threads()
    out = {"unit": "volume-pairs/sec", "kind": "port", "sample": "1 pair 64x64x64 forward, oracle/modet_torch.py (ATen-CPU fp32), best of 3"}
    for nt, key in ((min(cores, 32), "value"), (1, "value_1_thread")):
        torch.set_num_threads(nt)
        best = 1e9
        with torch.no_grad():
            orc.modet_forward(p, mov, fix, (8, 4, 2, 1, 1), 6, 1.0)
            for _ in range(3):
                t0 = time.perf_counter()
                orc.modet_forward(p, mov, fix, (8, 4, 2, 1, 1), 6, 1.0)
                best = min(best, time.perf_counter() - t0)
        out[key] = 1.0 / best
        if key == "value":
            out["cores"] = nt
    torch.set_num_threads(cores)
    return out


def cfg5_leg(dev, steps):
    """BASELINE.json configs[4] on this one GPU: Mindboggle-sized 160x192x224 volumes, bf16 storage / fp32 accumulate,
    2 pairs per GPU, full train step replayed as a hipGraph (the N-GPU form adds the overlapped all-reduce).  A side leg of
    the default line so that the driver's single command records it next to the fp32 headline."""
    from smilecode_amd import models, synth
    from smilecode_amd.engine import Trainer
    shape, batch = (160, 192, 224), 2
    model = models.ModeT(shape, head_dim=6, num_heads=[8, 4, 2, 1, 1], scale=1, act_dtype=torch.bfloat16).to(dev)
    models.load_numpy_weights(model, synth.make_weights(24))
    tr = Trainer(model)
    mov, fix = (torch.from_numpy(a).to(dev) for a in synth.make_pair(shape, 24, batch))
    graphed = True
    try:
        tr.capture(mov, fix)
    except Exception as e:                                       # noqa: BLE001 -- the leg must not take the headline down
        tr.release_graph()
        graphed = False
        log(f"[bench] cfg5 leg: hipGraph capture failed ({e!r}); eager steps")
    for _ in range(3):
        tr.train_step(mov, fix, epoch=0)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(steps):
        loss = tr.train_step(mov, fix, epoch=0)[0]
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / steps
    lv = float(loss)
    if not (lv == lv and abs(lv) < 1e6 and bool(torch.isfinite(tr.fp.flat).all())):      # a diverged run times a different computation
        raise RuntimeError(f"cfg5 leg diverged (loss {lv})")
    out = {"value": batch / dt, "unit": "volume-pairs/sec", "ms_per_step": dt * 1e3, "steps": steps, "dtype": "bf16",
           "hip_graph": graphed, "loss_after": lv,
           "workload": "ModeT 160x192x224 bf16 storage / fp32 accumulate (ConvInsBlock chains), batch=2/GPU, full train step "
                       "NCC+Grad3d fwd+bwd+Adam-amsgrad, 1 GPU (the 8-GPU form of configs[4] adds the overlapped all-reduce)"}
    del tr, model, mov, fix
    torch.cuda.empty_cache()
    return out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--workload", choices=["train", "fwd", "allreduce", "op"], default="train")
    ap.add_argument("--shape", default="160,192,160")
    ap.add_argument("--batch", type=int, default=1, help="volume pairs per rank per step")
    ap.add_argument("--dtype", choices=["f32", "bf16"], default="f32",
                    help="bf16 = BASELINE.json configs[4]: bf16 storage / fp32 accumulate in the ConvInsBlock chains (separate line; the headline is f32)")
    ap.add_argument("--overlap", action="store_true",
                    help="N>1: all-reduce the gradients in three buckets launched from backward hooks (eager steps, no hipGraph)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-extra", action="store_true", help="skip the cfg-2 (forward+warp) and cfg-1 (CPU 64^3) side legs")
    ap.add_argument("--graph", choices=["auto", "on", "off"], default="auto",
                    help="train workload: replay forward+backward as one hipGraph in the timed region (auto = if capture succeeds)")
    ap.add_argument("--breakdown", default="", help="write the per-op breakdown of one profiled step to this JSON file")
    args = ap.parse_args()
    shape = tuple(int(s) for s in args.shape.split(","))

    from smilecode_amd import models, ops, synth
    from smilecode_amd.engine import Trainer
    from smilecode_amd.parallel import init_from_env

    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        self_launch(args.gpus)                       # does not return
    rank, local, world = init_from_env()
    if world != args.gpus:
        log(f"[bench] ERROR: --gpus {args.gpus} but the launcher started WORLD_SIZE={world} ranks")
        sys.exit(2)
    if args.workload == "allreduce":
        return allreduce_only(args, rank, local, world)
    if args.workload == "op":
        return operator_only(args, rank, local, world)
    if world > 1 and dist.get_backend() == "nccl" and torch.cuda.device_count() < world:
        log(f"[bench] ERROR: {world} ranks but only {torch.cuda.device_count()} GPU(s) visible")
        sys.exit(2)
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)

    model = models.ModeT(shape, head_dim=6, num_heads=[8, 4, 2, 1, 1], scale=1,
                         act_dtype=torch.bfloat16 if args.dtype == "bf16" else torch.float32).to(dev)
    models.load_numpy_weights(model, synth.make_weights(24))
    trainer = Trainer(model, overlap_allreduce=args.overlap)
    mov, fix = synth.make_pair(shape, 24 + 2 * args.batch * rank, args.batch)      # per-rank pairs, weak scaling
    mov, fix = torch.from_numpy(mov).to(dev), torch.from_numpy(fix).to(dev)

    last = {}

    def step():
        if args.workload == "train":
            last["loss"] = trainer.train_step(mov, fix, epoch=0)[0]      # device scalar, read once after the timed region
        else:
            trainer.infer(mov, fix)

    def barrier():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    # warmup (untimed); the last (up to three) warmup steps are profiled per op to find the dominant kernel group.  The
    # per-family time is the MEDIAN over those steps: a single profiled step can contain a one-off stall (a 10 ms
    # hiccup inside a 35 us kernel was observed once) that would crown the wrong family.
    nprof = max(1, min(3, args.warmup))       # --warmup 0 still gets one untimed profiled step
    for _ in range(max(args.warmup - nprof, 0)):
        step()
    torch.cuda.synchronize()
    summaries = []
    for _ in range(nprof):
        tm = ops.KernelTimer()
        ops.set_kernel_timer(tm)
        step()
        torch.cuda.synchronize()
        ops.set_kernel_timer(None)
        summaries.append(tm.summary())
    breakdown = {}
    for k in (summaries[0] if summaries else {}):
        runs = sorted((sm[k] for sm in summaries if k in sm), key=lambda v: v["ms"])
        breakdown[k] = dict(runs[len(runs) // 2])            # the median run of this tag
    fams = {}
    for k, v in breakdown.items():
        d = fams.setdefault(family(k), {"calls": 0, "ms": 0.0, "flops": 0.0, "bytes": 0.0, "pflops": 0.0, "tags": set()})
        for f in ("calls", "ms", "flops", "bytes"):
            d[f] += v[f]
        d["pflops"] += v["flops"] * piece_products(k)
        d["tags"].add(k)
    dominant = max(fams, key=lambda k: fams[k]["ms"]) if fams else None
    if rank == 0:
        tot = sum(v["ms"] for v in breakdown.values())
        log(f"[bench] per-op breakdown of one {args.workload} step (HIP events, median of {nprof} profiled steps, sum {tot:.3f} ms):")
        for k, v in sorted(breakdown.items(), key=lambda kv: -kv[1]["ms"]):
            log(f"   {k:26s} calls {v['calls']:3d}  {v['ms']:8.3f} ms  {v['flops'] / v['ms'] / 1e9 if v['ms'] else 0:9.2f} TFLOP/s"
                f"  {v['bytes'] / v['ms'] / 1e6 if v['ms'] else 0:9.1f} GB/s(alg)")
        log("[bench] by kernel family:")
        for k, v in sorted(fams.items(), key=lambda kv: -kv[1]["ms"])[:12]:
            log(f"   {k:26s} calls {v['calls']:3d}  {v['ms']:8.3f} ms ({100 * v['ms'] / tot:4.1f} %)")
        if args.breakdown:
            with open(args.breakdown, "w") as f:
                json.dump({"ops": breakdown, "families": {k: {kk: vv for kk, vv in v.items() if kk != "tags"}
                                                          for k, v in fams.items()}}, f, indent=1, sort_keys=True)

    # host-side cost of one step (Python + autograd + ~500 launches), measured as the time to enqueue a step on an
    # idle GPU: the device runs behind the host as long as this stays below ms_per_step
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    step()
    host_ms = (time.perf_counter() - t0) * 1e3
    torch.cuda.synchronize()
    if rank == 0:
        log(f"[bench] host enqueue time of one step: {host_ms:.2f} ms")

    # hipGraph: forward + losses + backward + gradient packing captured once, replayed per step (engine.Trainer.capture);
    # the all-reduce and the Adam kernel stay eager.  A replayed graph cannot carry per-kernel events, so the roofline leg
    # below is measured over eager steps right after the timed region.
    def quick(n=3):
        """max over ranks of the wall time of n steps (untimed probe used to choose between graph replay and eager)"""
        barrier()
        t0 = time.perf_counter()
        for _ in range(n):
            step()
        barrier()
        t = torch.tensor([time.perf_counter() - t0], dtype=torch.float64, device=dev)
        if world > 1:
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
        return float(t.item()) / n

    graphed = False
    if args.workload == "train" and args.graph != "off":
        t_eager = quick()
        try:
            trainer.capture(mov, fix)
            graphed = True
        except Exception as e:
            trainer.release_graph()
            if args.graph == "on":
                raise
            log(f"[bench] hipGraph capture failed ({e!r}); timing the eager path")
        if world > 1:                       # every rank must take the same path -- agreed on BEFORE any further step: a
            flag = torch.tensor([1 if graphed else 0], device=dev)     # step holds a 4 MB gradient all-reduce, and a rank
            dist.all_reduce(flag, op=dist.ReduceOp.MIN)                 # that skipped it would pair this flag with it
            if graphed and int(flag.item()) == 0:
                trainer.release_graph()
                graphed = False
        if graphed:
            for _ in range(2):
                step()
            torch.cuda.synchronize()
        if graphed and args.graph == "auto":
            t_graph = quick()
            if t_graph > 1.05 * t_eager:     # never let the replay cost throughput (same decision on every rank: both are maxima)
                log(f"[bench] hipGraph replay is slower here ({t_graph * 1e3:.2f} vs {t_eager * 1e3:.2f} ms/step eager): using the eager path")
                trainer.release_graph()
                graphed = False
                torch.cuda.empty_cache()
                for _ in range(2):           # absorb the one-off cost of returning the graph's memory pool
                    step()
                torch.cuda.synchronize()
    host_graph_ms = None
    if graphed:
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        step()
        host_graph_ms = (time.perf_counter() - t0) * 1e3
        torch.cuda.synchronize()
        if rank == 0:
            log(f"[bench] hipGraph replay: host enqueue time of one step {host_graph_ms:.2f} ms")

    # timed region: exactly K steps.  Eager: only the dominant group carries events.  Graph: no events inside.
    tsel = ops.KernelTimer(select=set(fams[dominant]["tags"]) if dominant else set())
    barrier()
    if not graphed:
        ops.set_kernel_timer(tsel)
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
    barrier()
    dt = time.perf_counter() - t0
    ops.set_kernel_timer(None)
    # A timing of a broken computation is worthless (a network that went NaN skips most of its scatter-adds and runs ~13 %
    # faster): the loss of the last timed step and every parameter must be finite, or the run is refused
    loss_timed = float(last["loss"]) if "loss" in last else None
    if args.workload == "train" and not (loss_timed == loss_timed and abs(loss_timed) < 1e6
                                         and bool(torch.isfinite(trainer.fp.flat).all())):
        log(f"[bench] ERROR: the train step diverged inside the timed region (loss {loss_timed}); refusing to report a number")
        sys.exit(3)
    if world > 1:
        t = torch.tensor([dt], dtype=torch.float64, device=dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())
    roof_steps, dt_eager = args.steps, None
    if graphed:
        # roofline leg: the same K steps again, eager, with HIP events around every launch of the dominant kernel
        trainer.release_graph()
        for _ in range(2):
            step()
        barrier()
        ops.set_kernel_timer(tsel)
        t0 = time.perf_counter()
        for _ in range(roof_steps):
            step()
        barrier()
        dt_eager = time.perf_counter() - t0
        ops.set_kernel_timer(None)

    ar_probe = allreduce_probe(trainer.fp.grad, world) if world > 1 else None     # after the timed region, every rank
    extra = {}
    if world == 1 and args.workload == "train" and not args.no_extra:
        # cfg 2 (forward + final warp, no grad) on the same pair, and the cfg-1 CPU leg (64^3 forward of the oracle)
        for _ in range(3):
            trainer.infer(mov, fix)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(args.steps):
            trainer.infer(mov, fix)
        torch.cuda.synchronize()
        tf = (time.perf_counter() - t0) / args.steps
        extra["cfg2_forward_warp"] = {"value": args.batch / tf, "unit": "volume-pairs/sec", "ms_per_step": tf * 1e3,
                                      "workload": "ModeT LPBA %dx%dx%d fp32, batch=%d, forward+warp, no grad" % (*shape, args.batch)}
        if shape == (160, 192, 160) and args.dtype == "f32" and args.batch == 1:
            try:
                extra["cfg5_bf16_160x192x224_b2"] = cfg5_leg(dev, max(5, min(args.steps, 20)))
            except Exception as e:                                     # noqa: BLE001
                extra["cfg5_bf16_160x192x224_b2"] = {"value": None, "workload": f"failed: {e!r}"}
        if not args.no_cpu_baseline:
            try:
                extra["cfg1_cpu_forward_64"] = cpu_cfg1()
            except Exception as e:
                extra["cfg1_cpu_forward_64"] = {"value": None, "sample": f"failed: {e!r}"}
    if rank == 0:
        pairs = args.steps * args.batch * world
        roof = None
        if dominant:
            d = {"calls": 0, "ms": 0.0, "flops": 0.0, "bytes": 0.0}
            pfl = 0.0
            for k, v in tsel.summary().items():
                for f in d:
                    d[f] += v[f]
                pfl += v["flops"] * piece_products(k)
            sec = d["ms"] * 1e-3
            roof = roof_of(dominant, d["flops"], d["bytes"], sec, pfl)
            roof.update({"kernel": dominant, "launches": d["calls"], "launches_per_step": d["calls"] / roof_steps,
                         "avg_launch_ms": d["ms"] / d["calls"], "share_of_step": d["ms"] / ((dt_eager or dt) * 1e3),
                         "measured_over": ("%d eager steps right after the timed region (the timed region replays a hipGraph, "
                                           "which cannot carry per-kernel events)" % roof_steps) if graphed else "the timed region",
                         "note": "all launches of this kernel symbol in the K timed steps; algorithmic work summed per "
                                 "launch shape (DESIGN.md section 4)" + ("; warp_bwd_kernel is bound by the L2 float-atomic "
                                 "unit (its d_src scatter), not by HBM: DESIGN.md section 4" if dominant == "warp_bwd_kernel" else "")})
            # HBM bytes from the rocprofv3 --pmc passes of tools/refresh_profiles.sh: quoted only while the profile was taken on
            # THESE kernel sources (fingerprint of smilecode_amd/csrc), otherwise null + the reason
            pj = None
            pmc = os.path.join(ROOT, "profiles", "pmc_traffic.json")
            if os.path.exists(pmc):
                try:
                    pj_ = json.load(open(pmc))
                    have, now = pj_.get("csrc_sha16"), csrc_sha16()
                    if have == now:
                        pj = pj_
                        roof["traffic_source"] = ("profiles/pmc_traffic.json (rocprofv3 --pmc FETCH_SIZE x2 + WRITE_SIZE, summed over the "
                                                  "family's kernel symbols %s), csrc %s" % (list(FAMILY_SYMBOLS.get(dominant, (dominant,))), now))
                    else:
                        roof["traffic_source"] = ("profiles/pmc_traffic.json is stale: taken on csrc %s, this build is %s "
                                                  "(re-run tools/refresh_profiles.sh)" % (have, now))
                except Exception:
                    pass
            add_traffic(roof, pj, dominant, d["calls"] / roof_steps, d["bytes"] / roof_steps)
        # the same object for the six largest families of the profiled warm-up steps (median of those steps)
        tot_ms = sum(v["ms"] for v in fams.values()) or 1.0
        roof_top = []
        for k, v in sorted(fams.items(), key=lambda kv: -kv[1]["ms"])[:6]:
            r = roof_of(k, v["flops"], v["bytes"], v["ms"] * 1e-3, v.get("pflops"))
            r.pop("peak_note", None)
            r.update({"kernel": k, "ms_per_step": v["ms"], "launches_per_step": v["calls"], "share_of_kernel_time": v["ms"] / tot_ms})
            add_traffic(r, pj if dominant else None, k, v["calls"], v["bytes"])
            roof_top.append(r)
        out = {
            "metric": "volume-pairs/sec (%dx%dx%d) %s" % (*shape, "fwd+bwd" if args.workload == "train" else "fwd+warp"),
            "value": pairs / dt, "unit": "volume-pairs/sec", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": dt / args.steps * 1e3, "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": args.dtype, "data": "synthetic",
            "config": {"workload": ("ModeT %dx%dx%d %s, batch=%d/GPU, %s" % (
                *shape, "fp32" if args.dtype == "f32" else "bf16 storage / fp32 accumulate (ConvInsBlock chains), fp32 elsewhere", args.batch, "full train step NCC+Grad3d fwd+bwd+Adam-amsgrad" + (" + RCCL grad all-reduce" if world > 1 else "")
                if args.workload == "train" else "forward+warp")),
                "shape": list(shape), "global_batch": args.batch * world, "parallelism": f"dp{world}",
                "allreduce": "3 buckets, each launched after its stage of the backward (three hipGraph segments), overlapped" if args.overlap else "one flat all-reduce after backward"},
            "roofline": roof, "roofline_top": roof_top, "host_enqueue_ms_per_step": host_graph_ms if graphed else host_ms,
            "loss_after_timed_region": loss_timed,
            "hip_graph": graphed, "eager": {"host_enqueue_ms_per_step": host_ms,
                                            "ms_per_step": dt_eager / roof_steps * 1e3 if dt_eager else None},
            # proof of the N-rank run: what torch.distributed itself reports, and the step's one collective timed alone
            "backend": dist.get_backend() if world > 1 else None,
            "world_size": dist.get_world_size() if world > 1 else 1,
            "allreduce": ar_probe,
            "extra": extra,
        }
        if world == 1 and not args.no_cpu_baseline:
            try:
                out["cpu_baseline"] = cpu_baseline(shape, args.workload)
            except Exception as e:                                     # the GPU number must still be reported
                out["cpu_baseline"] = {"value": None, "unit": "volume-pairs/sec", "cores": torch.get_num_threads(),
                                       "kind": "port", "sample": f"failed: {e!r}"}
        print(json.dumps(out), flush=True)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
