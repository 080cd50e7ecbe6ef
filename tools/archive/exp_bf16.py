"""exploration: accuracy of the bf16-storage path (ops + end to end) against fp64 references"""
import os, sys, time
import numpy as np, torch, torch.nn.functional as F
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from smilecode_amd import models, ops, synth
from oracle import modet_torch as orc
torch.manual_seed(0)

def r16(t):
    return t.bfloat16().double()

for (B, D, H, W, Cin, Cout, inbf) in [(2, 20, 24, 28, 4, 8, False), (2, 20, 24, 28, 8, 8, True), (1, 9, 12, 10, 16, 32, True), (1, 9, 12, 10, 64, 128, True),
                                      (1, 9, 12, 10, 128, 128, True), (1, 33, 17, 40, 32, 32, False), (2, 16, 16, 16, 16, 16, True), (1, 5, 6, 7, 64, 64, False)]:
    x = torch.randn(B, Cin, D, H, W)
    w = torch.randn(Cout, Cin, 3, 3, 3) * (1.0 / np.sqrt(27 * Cin))
    b = torch.randn(Cout)
    ref = F.conv3d(r16(x), r16(w), b.double(), padding=1)
    xcl = x.permute(0, 2, 3, 4, 1).contiguous().cuda()
    xin = xcl.bfloat16() if inbf else xcl
    y, st = ops.conv3d_bf16_forward(xin, w.cuda(), b.cuda(), True)
    got = y.float().permute(0, 4, 1, 2, 3).double().cpu()
    e = (got - ref).abs()
    print(f"fwd {Cin}->{Cout} {D}x{H}x{W} inbf={inbf}: max err {float(e.max()):.3e} rel-to-bf16-ulp {float((e / (ref.abs() * 2**-8 + 1e-3)).max()):.2f} refmax {float(ref.abs().max()):.2f}")
    yn = ops._InstNormLReLUBF16.apply(y, st, 1e-5, False).permute(0, 4, 1, 2, 3).double().cpu()
    refn = F.leaky_relu(F.instance_norm(ref, eps=1e-5), 0.1)
    print(f"     IN(fwd) max err {float((yn - refn).abs().max()):.3e}")
    dy = torch.randn(B, Cout, D, H, W)
    refdx = torch.nn.grad.conv3d_input(x.shape, r16(w), r16(dy), padding=1)
    dycl = dy.permute(0, 2, 3, 4, 1).contiguous().cuda().bfloat16()
    for dxbf in ((True, False) if Cin % 8 == 0 else (False,)):
        dx = ops.conv3d_bf16_backward_data(dycl, w.cuda(), Cin, dxbf).float().permute(0, 4, 1, 2, 3).double().cpu()
        e = (dx - refdx).abs()
        print(f"     dgrad dxbf={dxbf}: max err {float(e.max()):.3e} vs-ulp {float((e / (refdx.abs() * 2**-8 + 1e-3)).max()):.2f}")

# end to end
for shape in ((32, 48, 32), (64, 64, 64)):
    w = synth.make_weights(24)
    mov_np, fix_np = synth.make_pair(shape, 24)
    p64 = {n: torch.from_numpy(v).double().requires_grad_(True) for n, v in w.items()}
    loss64, sim64, reg64, y64, f64 = orc.train_loss(p64, torch.from_numpy(mov_np).double(), torch.from_numpy(fix_np).double(), (8, 4, 2, 1, 1), 6, 1.0)
    g64 = dict(zip(p64, torch.autograd.grad(loss64, list(p64.values()))))
    from smilecode_amd import losses
    for dt in (torch.float32, torch.bfloat16):
        m = models.ModeT(shape, head_dim=6, num_heads=[8, 4, 2, 1, 1], scale=1, act_dtype=dt).cuda()
        models.load_numpy_weights(m, w)
        mov, fix = torch.from_numpy(mov_np).cuda(), torch.from_numpy(fix_np).cuda()
        y, flow = m(mov, fix)
        loss = losses.NCC_vxm()(fix, y) + losses.Grad3d(penalty="l2")(flow, fix)
        loss.backward()
        ef = (flow.double().cpu() - f64.detach()).abs()
        worst = 0.0
        for n, prm in m.named_parameters():
            gm = float(g64[n].abs().max())
            if gm < 1e-8: continue
            worst = max(worst, float((prm.grad.double().cpu() - g64[n]).abs().max()) / gm)
        print(f"e2e {shape} {dt}: flow max err {float(ef.max()):.3e} rms {float(ef.pow(2).mean().sqrt()):.3e} (|flow|max {float(f64.abs().max()):.1f}); loss err {abs(float(loss) - float(loss64)):.2e}; worst grad rel {worst:.3e}")
