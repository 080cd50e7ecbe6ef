"""conv3d_q.hip (bf16x3 forward / data gradient, K packed in channel quads) at the tail shapes of one train step:
accuracy against ATen-CPU fp64 (`check`: small ragged shapes, forward + fused statistics + normalised input + data gradient)
and median HIP-event time per call (`time`; a tuning build + MODET_CONV_Q=0 times the kernels it replaces)."""
import json, os, sys
import numpy as np
import torch
import torch.nn.functional as F
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from smilecode_amd import ops

L2, L3, L4, L5 = (80, 96, 80), (40, 48, 40), (20, 24, 20), (10, 12, 10)
LAYERS = [(16, 32, L3, 2), (32, 32, L3, 2), (32, 64, L4, 2), (64, 64, L4, 2), (64, 128, L5, 2), (128, 128, L5, 2),
          (6, 12, L2, 1), (12, 12, L2, 1), (12, 2, L2, 1), (12, 24, L3, 1), (24, 24, L3, 1), (24, 4, L3, 1),
          (24, 48, L4, 1), (48, 48, L4, 1), (48, 8, L4, 1)]


def timed(fn, iters=15):
    for _ in range(3):
        fn()
    ts = []
    for _ in range(iters):
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record(); fn(); b.record(); b.synchronize()
        ts.append(a.elapsed_time(b))
    ts.sort()
    return ts[len(ts) // 2]


def check():
    cases = [(16, 32, (5, 9, 20), 2), (32, 32, (4, 6, 9), 2), (12, 2, (6, 6, 18), 1), (6, 12, (7, 10, 13), 1), (12, 12, (9, 17, 8), 1),
             (24, 4, (5, 8, 16), 1), (24, 24, (4, 9, 11), 2), (48, 8, (3, 5, 10), 1), (24, 48, (4, 8, 8), 1), (64, 128, (2, 3, 10), 2),
             (128, 128, (3, 4, 5), 2), (2, 12, (6, 6, 18), 1), (4, 24, (5, 8, 16), 1), (8, 48, (3, 5, 10), 1), (20, 20, (5, 6, 7), 1),
             (64, 64, (4, 5, 6), 1)]
    for cin, cout, shape, B in cases:
        gen = torch.Generator().manual_seed(cin * 31 + cout)
        x = torch.randn((B, cin) + shape, generator=gen).double()
        w = (torch.randn((cout, cin, 3, 3, 3), generator=gen) / (27 * cin) ** 0.5).double()
        b = torch.randn(cout, generator=gen).double()
        gy = torch.randn((B, cout) + shape, generator=gen).double()
        y = F.conv3d(x, w, b, padding=1)
        dx = torch.nn.grad.conv3d_input(x.shape, w, gy, padding=1)
        cl = lambda t: t.permute(0, 2, 3, 4, 1).contiguous().float().cuda()
        xd, wd, bd, gd = cl(x), w.float().cuda(), b.float().cuda(), cl(gy)
        fam = ops._L().modet_conv3d_kernel_family(B, *shape, cin, cout, 0)
        fam_d = ops._L().modet_conv3d_kernel_family(B, *shape, cin, cout, 1)
        yh = ops.conv3d_forward(xd, wd, bd, False)
        dxh = ops.conv3d_backward_data(gd, wd, cin)
        e_y = float((yh.double().cpu().permute(0, 4, 1, 2, 3) - y).abs().max() / y.abs().max())
        e_dx = float((dxh.double().cpu().permute(0, 4, 1, 2, 3) - dx).abs().max() / dx.abs().max())
        msg = f"{cin:3d}->{cout:3d} {shape} B={B} family fwd {fam} dgrad {fam_d}: rel err y {e_y:.2e} d_x {e_dx:.2e}"
        if cout % 4 == 0:                               # ConvInsBlock: conv + fused statistics -> InstanceNorm + LeakyReLU
            z = ops.conv3d_instnorm_lrelu(xd, wd, bd)
            zr = F.leaky_relu(F.instance_norm(y, eps=1e-5), 0.1)
            e_z = float((z.double().cpu().permute(0, 4, 1, 2, 3) - zr).abs().max())
            msg += f" instnorm(y) {e_z:.2e}"
            assert e_z < 5e-5, msg
            if cin % 4 == 0:                            # inference path: lazily normalised input + statistics
                with torch.no_grad():
                    raw, st = ops.conv3d_with_stats(xd, wd, bd)
                    w2 = (torch.randn((cout, cout, 3, 3, 3), generator=gen) / (27 * cout) ** 0.5)
                    z2, _ = ops.lazy_instnorm_conv3d(raw, st, w2.cuda(), bd, want_stats=False)
                z2r = F.conv3d(zr, w2.double(), b, padding=1)
                e_z2 = float((z2.double().cpu().permute(0, 4, 1, 2, 3) - z2r).abs().max() / z2r.abs().max())
                msg += f" conv(norm(y)) {e_z2:.2e}"
                assert e_z2 < 5e-5, msg
        print(msg, flush=True)
        assert e_y < 2e-5 and e_dx < 2e-5, msg
        assert torch.equal(yh, ops.conv3d_forward(xd, wd, bd, False)), "not deterministic"


def time_all():
    tf, td = 0.0, 0.0
    for cin, cout, shape, B in LAYERS:
        g = torch.Generator(device="cuda").manual_seed(cin * 100 + cout)
        x = torch.randn((B,) + shape + (cin,), device="cuda", generator=g)
        dy = torch.randn((B,) + shape + (cout,), device="cuda", generator=g)
        w = torch.randn((cout, cin, 3, 3, 3), device="cuda", generator=g) / (27 * cin) ** 0.5
        b = torch.randn((cout,), device="cuda", generator=g)
        fam = ops._L().modet_conv3d_kernel_family(B, *shape, cin, cout, 0)
        fam_d = ops._L().modet_conv3d_kernel_family(B, *shape, cin, cout, 1)
        fwd = timed(lambda: ops.conv3d_forward(x, w, b, False))
        dg = timed(lambda: ops.conv3d_backward_data(dy, w, cin))
        fl = 54.0 * cin * cout * B * shape[0] * shape[1] * shape[2]
        tf += fwd; td += dg
        print("%-9s %-11s B=%d  fwd fam %d %.4f ms %6.1f TF/s   dgrad fam %d %.4f ms %6.1f TF/s" % (
            f"{cin}->{cout}", "x".join(map(str, shape)), B, fam, fwd, fl / fwd / 1e9, fam_d, dg, fl / dg / 1e9), flush=True)
    print("sum fwd %.4f ms, dgrad %.4f ms (each call includes its weight-packing launch)" % (tf, td))


if __name__ == "__main__":
    check() if sys.argv[1] == "check" else time_all()
