"""where does the gradient error of bf16 LEVEL FEATURES come from?  fp64-oracle-free check: bf16 model variants against the fp32 HIP run.
  A: features16 off (today's committed state)   B: features16 on (the real thing)
  C: features16 off + level features rounded by hand, INCLUDING what the pool reads (straight-through)"""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from smilecode_amd import models, ops, synth, losses

class _Round(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x): return x.bfloat16().float()
    @staticmethod
    def backward(ctx, g): return g

def run(shape, dtype, f16, emul=False, seed=24):
    m = models.ModeT(shape, head_dim=6, num_heads=[8, 4, 2, 1, 1], scale=1, act_dtype=dtype).cuda()
    m.encoder.features16 = f16
    models.load_numpy_weights(m, synth.make_weights(seed))
    orig = ops.conv_ins_pair_bf16_pool_split
    if emul:
        def patched(inp, w1, b1, w2, b2, Bh, eps=1e-5, features16=False):
            y = _Round.apply(ops.conv_ins_pair_bf16(inp, w1, b1, w2, b2, eps))
            return ops.pool_tee_split(y, Bh)
        ops.conv_ins_pair_bf16_pool_split = patched
    mov, fix = (torch.from_numpy(a).cuda() for a in synth.make_pair(shape, seed))
    y, flow = m(mov, fix)
    loss = losses.NCC_vxm()(fix, y) + losses.Grad3d("l2")(flow)
    g = torch.autograd.grad(loss, list(m.parameters()))
    ops.conv_ins_pair_bf16_pool_split = orig
    names = [n for n, _ in m.named_parameters()]
    return flow.detach().double(), [t.double() for t in g], names

for shape in ((64, 64, 64),):
    f0, g0, names = run(shape, torch.float32, False)
    for tag, kw in (("A features fp32", dict(f16=False)), ("B features16", dict(f16=True)), ("C emulated rounding incl. pool", dict(f16=False, emul=True))):
        f, g, _ = run(shape, torch.bfloat16, **kw)
        G, G0 = torch.cat([t.reshape(-1) for t in g]), torch.cat([t.reshape(-1) for t in g0])
        print(f"{shape} {tag}: flow rms {float((f - f0).pow(2).mean().sqrt()):.4f} grad relL2 {float((G - G0).norm() / G0.norm()):.4f}")
        if tag.startswith("B"):
            worst = sorted(((float((a - b).norm() / (G0.norm())), n) for a, b, n in zip(g, g0, names)), reverse=True)[:6]
            print("   largest contributions to the B error:", [(n, round(v, 4)) for v, n in worst])
