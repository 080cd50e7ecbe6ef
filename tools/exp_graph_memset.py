"""Is the hipMemsetAsync inside modet_warp_bwd replayed by a captured hipGraph?  warp backward captured alone, replayed 3x."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from smilecode_amd import _lib, ops
L = _lib.load()
B, D, H, W, C = 1, 16, 16, 16, 8
g = torch.Generator(device="cuda").manual_seed(0)
src = torch.randn((B, D, H, W, C), device="cuda", generator=g)
fl = 2.0 * torch.randn((B, D, H, W, 3), device="cuda", generator=g)
dout = torch.randn((B, D, H, W, C), device="cuda", generator=g)
dsrc, dflow = torch.empty_like(src), torch.empty_like(fl)
def run(stream):
    _lib.check(L.modet_warp_bwd(src.data_ptr(), fl.data_ptr(), dout.data_ptr(), dsrc.data_ptr(), dflow.data_ptr(), B, D, H, W, C, 0, 0, stream), "warp_bwd")
run(torch.cuda.current_stream().cuda_stream); torch.cuda.synchronize()
ref = dsrc.clone()
s = torch.cuda.Stream()
s.wait_stream(torch.cuda.current_stream())
with torch.cuda.stream(s):
    run(s.cuda_stream)
torch.cuda.current_stream().wait_stream(s)
gr = torch.cuda.CUDAGraph()
with torch.cuda.graph(gr):
    run(torch.cuda.current_stream().cuda_stream)
for rep in range(3):
    gr.replay(); torch.cuda.synchronize()
    print("replay %d: max |d_src - eager| = %.3e   (|d_src| max %.3e)" % (rep, float((dsrc - ref).abs().max()), float(ref.abs().max())))
