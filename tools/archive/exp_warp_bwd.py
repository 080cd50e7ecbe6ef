import torch, time, sys, os
sys.path.insert(0, os.getcwd())
from smilecode_amd import ops
torch.manual_seed(0)
B,D,H,W,C = 1,160,192,160,8
src = torch.randn(B,D,H,W,C, device='cuda')
dout = torch.randn(B,D,H,W,C, device='cuda')
for name, flow in (("smooth", torch.randn(B,D//16,H//16,W//16,3,device='cuda')), ("rough", None)):
    if flow is None:
        flow = 2*torch.randn(B,D,H,W,3,device='cuda')
    else:
        flow = 3*torch.nn.functional.interpolate(flow.permute(0,4,1,2,3), size=(D,H,W), mode='trilinear').permute(0,2,3,4,1).contiguous()
    src.requires_grad_(True); flow.requires_grad_(True)
    out = ops.warp(src, flow, 0, False)
    for _ in range(2): torch.autograd.grad(out, [src, flow], dout, retain_graph=True)
    torch.cuda.synchronize(); t=time.perf_counter()
    for _ in range(5): g = torch.autograd.grad(out, [src, flow], dout, retain_graph=True)
    torch.cuda.synchronize(); print(os.environ.get("MODET_WARP_BWD_VARIANT","0"), name, "warp_bwd C8 ms:", (time.perf_counter()-t)/5*1e3, float(g[0].abs().sum()))
