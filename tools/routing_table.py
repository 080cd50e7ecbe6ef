"""The conv routing of the train step as a markdown table: for every 3x3x3 conv layer of ModeT at a given
volume, which kernel family libmodet_hip.so runs for the forward, data-gradient and weight-gradient launch, and on how many
pieces.  Host-only (asks modet_conv3d_kernel_family_v; no GPU needed).

    python tools/routing_table.py [160,192,160]
"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from smilecode_amd import _lib  # noqa: E402

L = _lib.load()
shape = tuple(int(v) for v in (sys.argv[1] if len(sys.argv) > 1 else "160,192,160").split(","))
FAM = {0: "exact-f32 MFMA `conv3d_mfma_kernel`", 1: "tiled bf16x3 `conv3d_bf16_kernel<SP=3>`", 2: "z-march `conv_x3_kernel`",
       3: "direct exact-f32 `conv_direct_kernel`", 4: "`conv_wgrad_tr_kernel`", 5: "`conv_q_kernel`"}
WFAM = {0: "exact-f32 `conv3d_wgrad_kernel`", 2: "z-march `conv_x3_wgrad_kernel`", 4: "transpose-read `conv_wgrad_tr_kernel`"}


def lvl(k):
    return tuple(s >> k for s in shape)


rows = []
enc = [(1, 4), (4, 8), (8, 8), (8, 16), (16, 16), (16, 32), (32, 32), (32, 64), (64, 64), (64, 128), (128, 128)]
elv = [0, 0, 0, 1, 1, 2, 2, 3, 3, 4, 4]
# x_act: the input is a normalised activation (f16 forward form); variant: 1 = fused LeakyReLU, 3 = fused statistics, 2 = lazily normalised input
for i, ((ci, co), k) in enumerate(zip(enc, elv)):
    first = i == 0
    second_of_pair = i in (2, 4, 6, 8, 10)
    rows.append((f"encoder L{k + 1} {ci}->{co}", 2, lvl(k), ci, co, 1 if first else 3, not first and i != 1, second_of_pair))
for heads, k in ((8, 3), (4, 2), (2, 1)):       # CWM at the resolution of level k+1 (its output level): c = 3 heads
    c = 3 * heads
    d = lvl(k)
    rows.append((f"CWM{k + 2} {c}->{2 * c}", 1, d, c, 2 * c, 3, True, False))
    rows.append((f"CWM{k + 2} {2 * c}->{2 * c}", 1, d, 2 * c, 2 * c, 3, True, True))
    rows.append((f"CWM{k + 2} {2 * c}->{heads}", 1, d, 2 * c, heads, 0, True, True))
print("| layer | launch volume (B x D x H x W) | forward | data gradient | weight gradient |")
print("|---|---|---|---|---|")
for name, B, d, ci, co, var, x_act, lazy in rows:
    D, H, W = d
    ff = L.modet_conv3d_kernel_family_v(B, D, H, W, ci, co, 0, var)
    fd = L.modet_conv3d_kernel_family_v(B, D, H, W, ci, co, 1, 0)
    fw = L.modet_conv3d_kernel_family_v(B, D, H, W, ci, co, 2, 0)
    if ci == 1:
        fwd, dg, wg = "VALU z-march stencil `conv_c1_march_kernel` (+LeakyReLU)", "— (input needs no gradient)", "`conv_c1_wgrad_mfma_kernel` (exact f32, LeakyReLU' folded in)"
    else:
        pf = "2 f16 pieces" if (x_act and ff in (2, 5)) else ("3 bf16 pieces" if ff in (1, 2, 5) else "f32")
        fwd = f"{FAM[ff]}, {pf}" + (", fused IN statistics" if var == 3 else "")
        pd = "2 f16 pieces (max|d_y| from the IN backward)" if fd in (2, 5) else ("3 bf16 pieces" if fd == 1 else "f32")
        dg = f"{FAM[fd]}, {pd}"
        pw = ("2 f16 pieces" if x_act else "3 bf16 pieces") if fw in (2, 4) else "f32"
        wg = f"{WFAM.get(fw, FAM.get(fw, str(fw)))}, {pw}"
    print(f"| {name} | {B} x {D} x {H} x {W} | {fwd} | {dg} | {wg} |")
