"""time modet_warp_bwd at the level-1 feature-warp shape with only d_src, only d_flow, and both"""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from smilecode_amd import _lib, ops, synth
L = _lib.load()
shape = (160, 192, 160)
C = int(sys.argv[1]) if len(sys.argv) > 1 else 8
amp = float(sys.argv[2]) if len(sys.argv) > 2 else 3.0
fl = torch.from_numpy(synth.make_flow(shape, seed=3, amp=amp)).cuda().permute(0, 2, 3, 4, 1).contiguous()
src = torch.randn(1, *shape, C, device="cuda")
dout = torch.randn(1, *shape, C, device="cuda")
dsrc, dflow = torch.empty_like(src), torch.empty_like(fl)
st = torch.cuda.current_stream().cuda_stream
def run(ds, df):
    _lib.check(L.modet_warp_bwd(src.data_ptr(), fl.data_ptr(), dout.data_ptr(), ds, df, 1, *shape, C, 0, 0, st), "warp_bwd")
def t(ds, df, n=20):
    for _ in range(3): run(ds, df)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): run(ds, df)
    e1.record(); e1.synchronize()
    return e0.elapsed_time(e1) / n
print(f"C={C} amp={amp}: both {t(dsrc.data_ptr(), dflow.data_ptr()):.3f} ms, d_src only {t(dsrc.data_ptr(), None):.3f} ms, d_flow only {t(None, dflow.data_ptr()):.3f} ms")
with torch.no_grad():
    for _ in range(3): ops.warp(src, fl, 0, False)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(20): ops.warp(src, fl, 0, False)
    e1.record(); e1.synchronize()
    print(f"   warp_fwd {e0.elapsed_time(e1) / 20:.3f} ms")
