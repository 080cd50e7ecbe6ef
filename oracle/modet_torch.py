"""TEST INFRASTRUCTURE ONLY -- CPU oracle for the ModeT hot path (functional, ATen-CPU).

Only ``tests/``, ``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline`` leg
may import this file; the product package ``smilecode_amd`` never does.

A from-scratch functional restatement (pure functions over a name->tensor dict, no
nn.Module) of the reference forward path, differentiable through torch autograd on
CPU in fp32 or fp64.  The arithmetic primitives (conv3d, instance_norm, layer_norm,
avg_pool3d, trilinear interpolate) are ATen's -- the same third-party dependency the
reference calls (PyTorch, pinned only by README.md:16 "PyTorch
1.11"; semantics unchanged in 2.10).  The two custom pieces are restated directly:

* neighbourhood attention: 27 shifted dot products instead of the reference's
  unfold "memory boom" (ModeT/models.py:308-334), same math;
* warp: direct voxel-coordinate trilinear gather with zero padding instead of the
  normalise -> grid_sample(align_corners=True) round trip (ModeT/models.py:50-67),
  which is the identity in exact arithmetic.

Parity pin: checked against the imported reference in this build container by
tests/golden/make_goldens.py, whose outputs are committed under tests/golden/
(the reference ships no tests or vectors of its own: "parity unpinned" upstream).
"""
from __future__ import annotations

import math

import torch
import torch.nn.functional as F

LRELU = 0.1


# ----------------------------------------------------------------------------- encoder
def conv_block(p, name, x):
    """ConvBlock: conv3d(3,1,1) + LeakyReLU(0.1)  (ModeT/models.py:119-133)."""
    return F.leaky_relu(F.conv3d(x, p[name + ".main.weight"], p[name + ".main.bias"], padding=1), LRELU)


def conv_ins_block(p, name, x):
    """ConvInsBlock: conv3d + InstanceNorm3d(affine=False, eps=1e-5) + LeakyReLU(0.1)
    (ModeT/models.py:135-151)."""
    y = F.conv3d(x, p[name + ".main.weight"], p[name + ".main.bias"], padding=1)
    return F.leaky_relu(F.instance_norm(y, eps=1e-5), LRELU)


def encoder(p, x, taps=None, tag=""):
    """Five-level pyramid (ModeT/models.py:181-228).  ``taps`` receives every block output as ``enc{tag}.{lvl}.{i}``."""
    def tap(name, t):
        if taps is not None:
            taps[f"enc{tag}.{name}"] = t
        return t

    o = tap("0.0", conv_block(p, "encoder.conv0.0", x))
    o = tap("0.1", conv_ins_block(p, "encoder.conv0.1", o))
    o0 = tap("0.2", conv_ins_block(p, "encoder.conv0.2", o))
    outs = [o0]
    cur = o0
    for lvl in range(1, 5):
        cur = F.avg_pool3d(cur, 2)
        cur = tap(f"{lvl}.1", conv_ins_block(p, f"encoder.conv{lvl}.1", cur))
        cur = tap(f"{lvl}.2", conv_ins_block(p, f"encoder.conv{lvl}.2", cur))
        outs.append(cur)
    return outs


# ----------------------------------------------------------------------------- projection
def projection(p, name, feat):
    """(B,C,D,H,W) -> (B,D,H,W,dim): Linear + LayerNorm(eps 1e-5) (ModeT/models.py:230-241)."""
    f = feat.permute(0, 2, 3, 4, 1)
    y = F.linear(f, p[name + ".proj.weight"], p[name + ".proj.bias"])
    return F.layer_norm(y, (y.shape[-1],), p[name + ".norm.weight"], p[name + ".norm.bias"], 1e-5)


# ----------------------------------------------------------------------------- attention
def neighbourhood_logits(q, k, rpb, heads, scale):
    """q,k (B,D,H,W,heads*d) channels-last -> logits (B,heads,D,H,W,27).

    logit[t] = scale * q . k[n + off(t)] + rpb[h,t], t = 9*ki+3*kj+kk, off = (ki-1,kj-1,kk-1),
    out-of-volume k = 0 so the logit there is rpb[h,t] (ModeT/models.py:313-327;
    ModeT-cu/modet/modet_kernel.cu:44-83)."""
    B, D, H, W, C = q.shape
    d = C // heads
    qh = q.reshape(B, D, H, W, heads, d) * scale
    kp = F.pad(k.reshape(B, D, H, W, heads, d), (0, 0, 0, 0, 1, 1, 1, 1, 1, 1))
    cols = []
    for ki in range(3):
        for kj in range(3):
            for kk in range(3):
                ks = kp[:, ki:ki + D, kj:kj + H, kk:kk + W]
                cols.append((qh * ks).sum(-1))          # (B,D,H,W,heads)
    logits = torch.stack(cols, -1)                      # (B,D,H,W,heads,27)
    logits = logits + rpb.reshape(heads, 27)
    return logits.permute(0, 4, 1, 2, 3, 5)


def offsets27(dtype, device=None):
    """The 27 'values' of the attention: (ki-1,kj-1,kk-1) row-major (ModeT/models.py:293-301)."""
    r = torch.arange(-1, 2, dtype=dtype, device=device)
    g = torch.stack(torch.meshgrid(r, r, r, indexing="ij"), -1)
    return g.reshape(27, 3)


def mode_transformer(q, k, rpb, heads, scale):
    """-> (B, heads*3, D,H,W): softmax over the 27 modes, expected offset (ModeT/models.py:328-334)."""
    B, D, H, W, _ = q.shape
    attn = neighbourhood_logits(q, k, rpb, heads, scale).softmax(-1)
    x = attn @ offsets27(q.dtype, q.device)             # (B,heads,D,H,W,3)
    return x.permute(0, 1, 5, 2, 3, 4).reshape(B, heads * 3, D, H, W)


def correlation3d(mov, fix, kernel_size=3, d=3, sw=1, sf=2):
    """PR++ Correlation3D ("Baseline methods/PR++/models.py":205-232): NCDHW features -> (B, d^3, D,H,W).
    Box sums of both feature maps (grouped all-ones conv), then the channel dot product of mov's with fix's shifted
    by (i,j,k)*sf - sf voxels, / kernel_size^3."""
    B, C, H, W, T = mov.shape
    w = torch.ones((C, 1, kernel_size, kernel_size, kernel_size), dtype=mov.dtype)
    pm = F.conv3d(mov, w, stride=sw, padding=1, groups=C)
    pf = F.conv3d(fix, w, stride=sw, padding=sf + 1, groups=C)
    out = []
    for i in range(d):
        for j in range(d):
            for k in range(d):
                crop = pf[:, :, i * sf:i * sf + H, j * sf:j * sf + W, k * sf:k * sf + T]
                out.append((pm * crop).sum(1, keepdim=True))
    return torch.cat(out, 1) / kernel_size ** 3


# ----------------------------------------------------------------------------- warp
def warp(src, flow, mode="bilinear"):
    """out[b,c,p] = sample(src[b,c], p + flow[b,:,p]), zero padding
    (SpatialTransformer, ModeT/models.py:25-67).  Direct coordinates."""
    B, C, D, H, W = src.shape
    dev, dt = src.device, src.dtype
    gz = torch.arange(D, dtype=dt, device=dev).view(1, D, 1, 1)
    gy = torch.arange(H, dtype=dt, device=dev).view(1, 1, H, 1)
    gx = torch.arange(W, dtype=dt, device=dev).view(1, 1, 1, W)
    z = gz + flow[:, 0]
    y = gy + flow[:, 1]
    x = gx + flow[:, 2]
    flat = src.reshape(B, C, D * H * W)

    def fetch(iz, iy, ix):
        ok = (iz >= 0) & (iz < D) & (iy >= 0) & (iy < H) & (ix >= 0) & (ix < W)
        lin = (iz.clamp(0, D - 1) * H + iy.clamp(0, H - 1)) * W + ix.clamp(0, W - 1)
        v = torch.gather(flat, 2, lin.reshape(B, 1, -1).expand(B, C, -1)).reshape(B, C, D, H, W)
        return v * ok.unsqueeze(1).to(dt)

    if mode == "nearest":
        # ATen grid_sampler nearest = nearbyint (round half to even)
        return fetch(torch.round(z).long(), torch.round(y).long(), torch.round(x).long())
    z0, y0, x0 = torch.floor(z), torch.floor(y), torch.floor(x)
    fz, fy, fx = (z - z0).unsqueeze(1), (y - y0).unsqueeze(1), (x - x0).unsqueeze(1)
    z0, y0, x0 = z0.long(), y0.long(), x0.long()
    out = 0
    for dz in (0, 1):
        wz = fz if dz else 1 - fz
        for dy in (0, 1):
            wy = fy if dy else 1 - fy
            for dx in (0, 1):
                wx = fx if dx else 1 - fx
                out = out + fetch(z0 + dz, y0 + dy, x0 + dx) * (wz * wy * wx)
    return out


def upsample2(x):
    """nn.Upsample(scale_factor=2, 'trilinear', align_corners=True) (ModeT/models.py:354)."""
    return F.interpolate(x, scale_factor=2, mode="trilinear", align_corners=True)


# ----------------------------------------------------------------------------- CWM
def cwm(p, name, x, heads):
    """Competitive weighting module (ModeT/models.py:243-275)."""
    x = upsample2(x)
    h = conv_ins_block(p, name + ".conv.0", x)
    h = conv_ins_block(p, name + ".conv.1", h)
    h = F.conv3d(h, p[name + ".conv.2.weight"], p[name + ".conv.2.bias"], padding=1)
    wgt = h.softmax(1)
    B, _, D, H, W = x.shape
    xs = x.reshape(B, heads, 3, D, H, W)
    return 2 * (xs * wgt.unsqueeze(2)).sum(1)


# ----------------------------------------------------------------------------- model
def modet_forward(p, moving, fixed, num_heads=(8, 4, 2, 1, 1), head_dim=6, scale=1.0, taps=None):
    """ModeT.forward (ModeT/models.py:377-412).  ``scale=None`` -> head_dim**-0.5 (:285).
    ``taps`` (dict) receives named intermediates for per-stage parity tests."""
    sc = scale if scale else head_dim ** -0.5
    M = encoder(p, moving, taps, "M")
    Fx = encoder(p, fixed, taps, "F")

    def level(lvl, Ff, Mf):
        heads = num_heads[5 - lvl]
        q = projection(p, f"projblock{lvl}", Ff)
        k = projection(p, f"projblock{lvl}", Mf)
        w = mode_transformer(q, k, p[f"mdt{lvl}.rpb"], heads, sc)
        if taps is not None:
            taps[f"q{lvl}"], taps[f"k{lvl}"], taps[f"mdt{lvl}"] = q, k, w
        if lvl >= 3:
            w = cwm(p, f"cwm{lvl}", w, heads)
        if taps is not None:
            taps[f"w{lvl}"] = w
        return w

    def tap(name, t):
        if taps is not None:
            taps[name] = t
        return t

    flow = tap("flow5", level(5, Fx[4], M[4]))
    w = level(4, Fx[3], tap("Mw4", warp(M[3], flow)))
    flow = tap("flow4", warp(upsample2(2 * flow), w) + w)
    w = level(3, Fx[2], tap("Mw3", warp(M[2], flow)))
    flow = tap("flow3", warp(upsample2(2 * flow), w) + w)
    w = level(2, Fx[1], tap("Mw2", warp(M[1], flow)))
    flow = tap("flow2", upsample2(2 * (warp(flow, w) + w)))
    w = level(1, Fx[0], tap("Mw1", warp(M[0], flow)))
    flow = warp(flow, w) + w
    y_moved = warp(moving, flow)
    if taps is not None:
        for i in range(5):
            taps[f"M{i + 1}"], taps[f"F{i + 1}"] = M[i], Fx[i]
    return y_moved, flow


# ----------------------------------------------------------------------------- losses
def ncc_loss(y_true, y_pred, win=9):
    """NCC_vxm (ModeT/losses.py:34-94): zero-padded box sums, -mean(cc).  win = an int (cubic) or [wz, wy, wx]; as the reference
    (losses.py:57) EVERY axis is padded by floor(win[0] / 2), whatever the other two sizes are."""
    Ii, Ji = y_true, y_pred
    w = [int(win)] * 3 if isinstance(win, int) else [int(v) for v in win]
    filt = torch.ones(1, 1, *w, dtype=Ii.dtype, device=Ii.device)
    pad = w[0] // 2

    def box(t):
        return F.conv3d(t, filt, padding=pad)

    I_sum, J_sum = box(Ii), box(Ji)
    I2_sum, J2_sum, IJ_sum = box(Ii * Ii), box(Ji * Ji), box(Ii * Ji)
    n = float(w[0] * w[1] * w[2])
    u_I, u_J = I_sum / n, J_sum / n
    cross = IJ_sum - u_J * I_sum - u_I * J_sum + u_I * u_J * n
    I_var = I2_sum - 2 * u_I * I_sum + u_I * u_I * n
    J_var = J2_sum - 2 * u_J * J_sum + u_J * u_J * n
    cc = cross * cross / (I_var * J_var + 1e-5)
    return -cc.mean()


def grad3d_loss(flow, penalty="l2"):
    """Grad3d (ModeT/losses.py:6-31): mean forward differences along the 3 axes / 3."""
    dy = (flow[:, :, 1:] - flow[:, :, :-1]).abs()
    dx = (flow[:, :, :, 1:] - flow[:, :, :, :-1]).abs()
    dz = (flow[:, :, :, :, 1:] - flow[:, :, :, :, :-1]).abs()
    if penalty == "l2":
        dy, dx, dz = dy * dy, dx * dx, dz * dz
    return (dx.mean() + dy.mean() + dz.mean()) / 3.0


def train_loss(p, moving, fixed, num_heads=(8, 4, 2, 1, 1), head_dim=6, scale=1.0, weights=(1.0, 1.0)):
    """loss of one training iteration (ModeT/train.py:122-129)."""
    y_moved, flow = modet_forward(p, moving, fixed, num_heads, head_dim, scale)
    sim = ncc_loss(fixed, y_moved)
    reg = grad3d_loss(flow)
    return weights[0] * sim + weights[1] * reg, sim, reg, y_moved, flow


# ----------------------------------------------------------------------------- optimiser
def adam_amsgrad_step(params, grads, state, lr, step, beta1=0.9, beta2=0.999, eps=1e-8):
    """torch.optim.Adam(amsgrad=True, weight_decay=0) single-tensor update (train.py:101).
    ``state`` = dict name -> (m, v, vmax); ``step`` counts from 1."""
    bc1 = 1 - beta1 ** step
    bc2 = 1 - beta2 ** step
    for n, w in params.items():
        g = grads[n]
        m, v, vmax = state[n]
        m.mul_(beta1).add_(g, alpha=1 - beta1)
        v.mul_(beta2).addcmul_(g, g, value=1 - beta2)
        torch.maximum(vmax, v, out=vmax)
        denom = (vmax.sqrt() / math.sqrt(bc2)).add_(eps)
        w.addcdiv_(m, denom, value=-lr / bc1)


def poly_lr(epoch, max_epoch=30, init_lr=1e-4, power=0.9):
    """adjust_learning_rate (ModeT/train.py:166-168)."""
    return round(init_lr * (1 - epoch / max_epoch) ** power, 8)


# ----------------------------------------------------------------------------- eval
def dice_voi(pred, true, nlabels=54):
    """dice_val_VOI (ModeT/utils.py:86-106): first batch element, labels 1..54."""
    pr, tr = pred[0, 0], true[0, 0]
    tot = 0.0
    for i in range(1, nlabels + 1):
        a, b = pr == i, tr == i
        inter = float((a & b).sum())
        union = float(a.sum()) + float(b.sum())
        tot += 2.0 * inter / (union + 1e-5)
    return tot / nlabels


def jacobian_determinant(disp):
    """jacobian_determinant_vxm (ModeT/utils.py:108-150) for a (3,D,H,W) displacement field -> (D,H,W) float64.

    The reference adds the int64 identity grid to the float32 field (numpy promotes to float64, utils.py:126-130), takes
    np.gradient (central differences (f[i+1]-f[i-1])/2 inside, one-sided f[1]-f[0] / f[n-1]-f[n-2] at the ends) and
    expands the 3x3 determinant along its first row (utils.py:139-144).  Same fp64 operations in the same order here,
    written with explicit slices instead of np.gradient, so the result is bit-identical and `det <= 0` counts are exact."""
    import numpy as np
    disp = np.asarray(disp)
    D, H, W = disp.shape[1:]
    f = disp.transpose(1, 2, 3, 0).astype(np.float64)
    grid = np.stack(np.meshgrid(np.arange(D), np.arange(H), np.arange(W), indexing="ij"), 3)
    f = f + grid

    def grad(ax):
        g = np.empty_like(f)
        n = f.shape[ax]
        ix = [slice(None)] * 4

        def s(a, b=None):
            j = list(ix)
            j[ax] = slice(a, b) if b is not None or a != -1 else -1
            return tuple(j)

        def at(i):
            j = list(ix)
            j[ax] = i
            return tuple(j)

        if n > 2:
            g[s(1, -1)] = (f[s(2, None)] - f[s(None, -2)]) / 2.0
        g[at(0)] = (f[at(1)] - f[at(0)]) / 1.0
        g[at(n - 1)] = (f[at(n - 1)] - f[at(n - 2)]) / 1.0
        return g

    dx, dy, dz = grad(0), grad(1), grad(2)
    d0 = dx[..., 0] * (dy[..., 1] * dz[..., 2] - dy[..., 2] * dz[..., 1])
    d1 = dx[..., 1] * (dy[..., 0] * dz[..., 2] - dy[..., 2] * dz[..., 0])
    d2 = dx[..., 2] * (dy[..., 0] * dz[..., 1] - dy[..., 1] * dz[..., 0])
    return d0 - d1 + d2
