"""A/B timing of the bf16 conv kernels at the level-1/2 shapes (median of N launches, HIP events); MODET_HIP_LIB selects the build"""
import json, os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from smilecode_amd import _lib, ops
from tools.ab_kernels import timeit
torch.manual_seed(0)
out = {"lib": os.path.basename(_lib.LIB_PATH)}
L1, L2, L3 = (2, 160, 192, 160), (2, 80, 96, 80), (2, 40, 48, 40)
with torch.no_grad():
    for shape, cin, cout, inbf in ((L1, 4, 8, False), (L1, 8, 8, True), (L2, 8, 16, False), (L2, 16, 16, True), (L3, 16, 32, False), (L3, 32, 32, True)):
        x = torch.randn(*shape, cin, device="cuda")
        x = x.bfloat16() if inbf else x
        w, b = torch.randn(cout, cin, 3, 3, 3, device="cuda") * 0.1, torch.randn(cout, device="cuda")
        dy = torch.randn(*shape, cout, device="cuda").bfloat16()
        tag = f"[{cin}->{cout}]@{shape[1]}"
        out["fwd" + tag] = timeit(lambda: ops.conv3d_bf16_forward(x, w, b, True), 20)
        out["dgrad" + tag] = timeit(lambda: ops.conv3d_bf16_backward_data(dy, w, cin, inbf), 20)
        out["wgrad" + tag] = timeit(lambda: ops.conv3d_bf16_backward_weight(x, dy), 20)
    x = torch.randn(*L1, 8, device="cuda").bfloat16()
    st = torch.cat([torch.zeros(16, device="cuda"), torch.ones(2 * 1 * 8 * 2, device="cuda"), torch.zeros(2 * 64 * 8 * 2, device="cuda")])
    out["in_apply_bf16[C8]@160"] = timeit(lambda: ops._InstNormLReLUBF16.apply(x, st, 1e-5, True), 20)
xr = torch.randn(*L1, 8, device="cuda").bfloat16().requires_grad_(True)
y = ops._InstNormLReLUBF16.apply(xr, st, 1e-5, True)
g = torch.randn(*L1, 8, device="cuda").bfloat16()
out["in_bwd_bf16[C8]@160"] = timeit(lambda: torch.autograd.grad(y, xr, g, retain_graph=True), 20)
print(json.dumps(out))
