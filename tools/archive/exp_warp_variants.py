"""time modet_warp_bwd of whatever library MODET_HIP_LIB names, on the flow tools/exp_warp_real.py captured"""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from smilecode_amd import _lib
L = _lib.load()
d = torch.load("/tmp/warp_real_C8.pt")
src, fl = d["src"].cuda(), d["flow"].cuda()
B, D, H, W, C = src.shape
dout = torch.randn_like(src); dsrc = torch.empty_like(src); dflow = torch.empty_like(fl)
st = torch.cuda.current_stream().cuda_stream
def run(ds, df):
    _lib.check(L.modet_warp_bwd(src.data_ptr(), fl.data_ptr(), dout.data_ptr(), ds, df, B, D, H, W, C, 0, 0, st), "warp_bwd")
def t(ds, df, n=20):
    for _ in range(3): run(ds, df)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): run(ds, df)
    e1.record(); e1.synchronize()
    return e0.elapsed_time(e1) / n
print(os.path.basename(_lib.LIB_PATH), f"both {t(dsrc.data_ptr(), dflow.data_ptr()):.3f} d_src only {t(dsrc.data_ptr(), None):.3f} d_flow only {t(None, dflow.data_ptr()):.3f}")
