"""Cold-process loop over tests/dp_worker.py (2 ranks on ONE GPU over gloo) in the four (overlap, graph) forms, against the
single-process batch-2 eager step: the error PER PARAMETER TENSOR, so a wrong stage / layer is named.

    python tools/repro_dp.py [--iters N] [--modes 11,01,10,00] [--shape 32,48,32]
"""
import argparse
import os
import subprocess
import sys
import tempfile
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from smilecode_amd import models, synth                      # noqa: E402
from smilecode_amd.engine import Trainer                      # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--iters", type=int, default=6)
ap.add_argument("--modes", default="11,01,10,00")
ap.add_argument("--shape", default="32,48,32")
ap.add_argument("--extra-env", default="")
ap.add_argument("--diag", action="store_true", help="workers save their LOCAL gradients, eager and replayed (capture is then not verified)")
ap.add_argument("--load", type=int, default=0, help="CPU burner processes beside the runs (the suite's oracle jobs saturate the host)")
args = ap.parse_args()
shape = tuple(int(s) for s in args.shape.split(","))


def reference():
    model = models.ModeT(shape, head_dim=6, num_heads=[8, 4, 2, 1, 1], scale=1).cuda()
    models.load_numpy_weights(model, synth.make_weights(24))
    tr = Trainer(model)
    mov, fix = synth.make_pair(shape, 24, 2)
    tr.train_step(torch.from_numpy(mov).cuda(), torch.from_numpy(fix).cuda(), epoch=0)
    torch.cuda.synchronize()
    names = [n for n, _ in model.named_parameters()]
    return tr.fp.grad.cpu().numpy(), names, tr.fp.offsets


def reference_rank(r):
    """rank r's own pair through a plain eager single-process step: what its LOCAL gradient must be"""
    model = models.ModeT(shape, head_dim=6, num_heads=[8, 4, 2, 1, 1], scale=1).cuda()
    models.load_numpy_weights(model, synth.make_weights(24))
    tr = Trainer(model)
    mov, fix = synth.make_pair(shape, 24, 2)
    tr._fwd_bwd(torch.from_numpy(mov[r:r + 1]).cuda(), torch.from_numpy(fix[r:r + 1]).cuda())
    torch.cuda.synchronize()
    return tr.fp.grad.cpu().numpy()


g1, names, offsets = reference()
g1b, _, _ = reference()
local = [reference_rank(0), reference_rank(1)] if args.diag else None
gmax = np.abs(g1).max()
print("reference batch-2 eager step: run-to-run max|diff|/max|g| = %.3e" % (np.abs(g1 - g1b).max() / gmax), flush=True)


def per_param(g):
    rows = []
    for n, (off, k) in zip(names, offsets):
        a, b = g1[off:off + k], g[off:off + k]
        d = np.abs(a - b).max()
        rows.append((d / gmax, d / max(np.abs(a).max(), 1e-30), n, k))
    rows.sort(reverse=True)
    return rows


extra = dict(kv.split("=", 1) for kv in args.extra_env.split(",") if kv)
burners = [subprocess.Popen([sys.executable, "-c", "import torch\ntorch.set_num_threads(8)\na = torch.randn(1536, 1536)\nwhile True: a = (a @ a).tanh()"])
           for _ in range(args.load)]
worst = {}
for it in range(args.iters):
    for mode in args.modes.split(","):
        overlap, graph = mode[0], mode[1]
        tmp = tempfile.mkdtemp(prefix="repro_dp_")
        env = dict(os.environ, MODET_DIST_BACKEND="gloo", HSA_ENABLE_IPC_MODE_LEGACY="0", MODET_OVERLAP=overlap,
                   MODET_GRAPH=graph, MODET_DP_DIAG="1" if args.diag else "0", **extra)
        port = 29300 + (os.getpid() + it * 7 + int(mode, 2)) % 400
        cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr",
               "127.0.0.1", "--master-port", str(port), os.path.join(ROOT, "tests", "dp_worker.py"), tmp,
               ",".join(map(str, shape))]
        t0 = time.time()
        r = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=400)
        dt = time.time() - t0
        if r.returncode != 0:
            print("iter %d mode %s: rc %d\n%s" % (it, mode, r.returncode, r.stderr[-1500:]), flush=True)
            continue
        g = np.load(os.path.join(tmp, "rank0.npz"))["grad"]
        rows = per_param(g)
        e = rows[0][0]
        worst[mode] = max(worst.get(mode, 0.0), e)
        print("iter %d mode %s (%.0f s): max|diff|/max|g| = %.3e  worst tensors: %s" % (
            it, mode, dt, e, "; ".join("%s %.2e (own %.2e)" % (n, a, b) for a, b, n, k in rows[:3])), flush=True)
        if e > 5e-6:
            for a, b, n, k in rows[:40]:
                if a > 2e-6:
                    print("      %-36s %8d  %.3e of global max, %.3e of own max" % (n, k, a, b), flush=True)
            for r in range(2):                       # which rank, and eager or replayed (MODET_DP_DIAG=1)
                z = np.load(os.path.join(tmp, "rank%d.npz" % r))
                for key in ("eager_local", "replay0_local", "replay1_local"):
                    if key in z:
                        d = np.abs(z[key] - local[r])
                        j = int(np.argmax(d))
                        seg = [n for n, (off, k) in zip(names, offsets) if off <= j < off + k][0]
                        print("      rank %d %-14s vs its plain eager step: %.3e of max|g| (in %s)" % (
                            r, key, d.max() / np.abs(local[r]).max(), seg), flush=True)
for b in burners:
    b.kill()
print("worst per mode:", {m: "%.3e" % v for m, v in worst.items()})
