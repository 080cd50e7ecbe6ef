"""warp backward, patch form with the y merge (warp_bwd2_kernel) against the x / z-merge form (warp_bwd_kernel), on the MODEL'S OWN
flows captured from a forward pass (every warp of a train step that scatters): result agreement and time.  Needs a tuning
build of warp.hip for the A/B switch:  bash tools/build_variant.sh warpT warp.hip "-DMODET_TUNING"
    MODET_HIP_LIB=build/variants/libmodet_hip_warpT.so python tools/exp_warp_bwd2.py"""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from smilecode_amd import _lib, models, ops, synth
L = _lib.load()
shape = tuple(int(v) for v in (sys.argv[1] if len(sys.argv) > 1 else "160,192,160").split(","))
m = models.ModeT(shape, head_dim=6, num_heads=[8, 4, 2, 1, 1], scale=1).cuda().eval()
models.load_numpy_weights(m, synth.make_weights(24))
mov, fix = (torch.from_numpy(a).cuda() for a in synth.make_pair(shape, 24))
rec = []
orig = ops.warp
def spy(src, flow, mode=0, add_flow=False, flow_bound=0):
    if not flow_bound:
        rec.append((src.detach().clone(), flow.detach().clone(), bool(add_flow)))
    return orig(src, flow, mode, add_flow, flow_bound)
ops.warp = spy
with torch.no_grad():
    m(mov, fix)
ops.warp = orig
st = torch.cuda.current_stream().cuda_stream
tot = {"0": 0.0, "1": 0.0}
for src, fl, addf in rec:
    B, D, H, W, C = src.shape
    dout = torch.randn_like(src)
    out = {}
    for mode in ("0", "1"):
        os.environ["MODET_WARP_BWD2"] = mode
        dsrc, dflow = torch.empty_like(src), torch.empty_like(fl)
        def run():
            _lib.check(L.modet_warp_bwd(src.data_ptr(), fl.data_ptr(), dout.data_ptr(), dsrc.data_ptr(), dflow.data_ptr(), B, D, H, W, C, int(addf), 0, st), "warp_bwd")
        for _ in range(3): run()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(20): run()
        e1.record(); e1.synchronize()
        out[mode] = (dsrc.clone(), dflow.clone(), e0.elapsed_time(e1) / 20)
        tot[mode] += out[mode][2]
    es = float((out["0"][0] - out["1"][0]).abs().max()) / max(1e-30, float(out["0"][0].abs().max()))
    ef = float((out["0"][1] - out["1"][1]).abs().max()) / max(1e-30, float(out["0"][1].abs().max()))
    print(f"C={C:3d} {D}x{H}x{W} add_flow={int(addf)}: x/z merges {out['0'][2]:.3f} ms, patch form {out['1'][2]:.3f} ms; rel diff d_src {es:.1e} d_flow {ef:.1e}")
print("sum: %.3f -> %.3f ms" % (tot["0"], tot["1"]))
