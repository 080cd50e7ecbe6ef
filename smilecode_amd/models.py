"""Drop-in ``ModeT`` / ``ModeT_cu`` modules for the MI355X-native hot path.

Same constructor signature, attribute names, ``forward(moving, fixed) -> (y_moved, flow)``
contract and state_dict keys as the reference (ModeT/models.py:338-412, ModeT-cu/models.py:
319-393), so train.py / infer.py / existing checkpoints work unchanged.  What differs is
underneath: every op is a hand-written HIP kernel behind libmodet_hip.so (smilecode_amd.ops),
activations are channels-last, moving+fixed go through the shared encoder as one batch, the
neighbourhood attention is a single fused kernel, and the SpatialTransformer computes its
identity grid in-kernel instead of keeping a 59 MB buffer per level.

state_dict compatibility:
  * parameters: identical names/shapes (``encoder.conv0.0.main.weight`` ... ``cwm5.conv.2.bias``);
  * ``mdtN.grid`` (3,3,3,3) [ModeT] / ``mdtN.v`` (27,3) [ModeT_cu]: kept as buffers, either flavour loads;
  * ``transformer.N.grid``: accepted and dropped on load; not emitted on save unless
    ``legacy_grid_buffers=True`` (then regenerated so the reference can load our checkpoints strictly).
"""
from __future__ import annotations

import math

import torch
import torch.nn as nn

from . import ops


class _SplitBatch(torch.autograd.Function):
    """(2B, ...) -> two (B, ...) views; backward writes both gradients straight into one buffer
    (plain slicing would zero-fill two full-size tensors and add them)."""

    @staticmethod
    def forward(ctx, x, B):
        ctx.B = B
        return x[:B], x[B:]

    @staticmethod
    def backward(ctx, ga, gb):
        B = ctx.B
        ref = ga if ga is not None else gb
        out = torch.empty((2 * B,) + tuple(ref.shape[1:]), dtype=ref.dtype, device=ref.device)
        if ga is None:
            out[:B].zero_()
        else:
            out[:B].copy_(ga)
        if gb is None:
            out[B:].zero_()
        else:
            out[B:].copy_(gb)
        return out, None


class SpatialTransformer(nn.Module):
    """N-D spatial transformer (reference ModeT/models.py:25-67), NCDHW in / NCDHW out.

    ``forward(src, flow)``: src (B,C,D,H,W), flow (B,3,D,H,W) in voxels.  mode 'bilinear' | 'nearest'."""

    def __init__(self, size, mode="bilinear", legacy_grid_buffers=False):
        super().__init__()
        self.size = tuple(int(s) for s in size)
        self.mode = mode
        self.legacy_grid_buffers = legacy_grid_buffers

    def _load_from_state_dict(self, state_dict, prefix, *args, **kwargs):
        state_dict.pop(prefix + "grid", None)          # reference checkpoints carry it (models.py:47)
        super()._load_from_state_dict(state_dict, prefix, *args, **kwargs)

    def _save_to_state_dict(self, destination, prefix, keep_vars):
        super()._save_to_state_dict(destination, prefix, keep_vars)
        if self.legacy_grid_buffers:
            vec = [torch.arange(0, s) for s in self.size]
            grid = torch.stack(torch.meshgrid(vec, indexing="ij")).unsqueeze(0).float()
            destination[prefix + "grid"] = grid

    def forward(self, src, flow):
        out = ops.warp(ops.to_channels_last(src.contiguous()), ops.to_channels_last(flow.contiguous()),
                       0 if self.mode == "bilinear" else 1, False)
        return ops.to_ncdhw(out)

    def forward_cl(self, src_cl, flow_cl, add_flow=False, flow_bound=0):
        return ops.warp(src_cl, flow_cl, 0 if self.mode == "bilinear" else 1, add_flow, flow_bound)


class _Conv3dParams(nn.Module):
    """parameter holder with nn.Conv3d's names, shapes and default init (weight (Cout,Cin,3,3,3), bias)."""

    def __init__(self, cin, cout):
        super().__init__()
        self.weight = nn.Parameter(torch.empty(cout, cin, 3, 3, 3))
        self.bias = nn.Parameter(torch.empty(cout))
        nn.init.kaiming_uniform_(self.weight, a=math.sqrt(5))
        bound = 1.0 / math.sqrt(cin * 27)
        nn.init.uniform_(self.bias, -bound, bound)


class ConvBlock(nn.Module):
    """conv + LeakyReLU(0.1) (reference models.py:119-133); channels-last in/out."""

    def __init__(self, in_channels, out_channels):
        super().__init__()
        self.main = _Conv3dParams(in_channels, out_channels)

    def forward(self, x):
        return ops.conv3d(x, self.main.weight, self.main.bias, True)


class ConvInsBlock(nn.Module):
    """conv + InstanceNorm3d + LeakyReLU(0.1) (reference models.py:135-151); channels-last in/out."""

    def __init__(self, in_channels, out_channels):
        super().__init__()
        self.main = _Conv3dParams(in_channels, out_channels)

    def forward(self, x):
        return ops.conv3d_instnorm_lrelu(x, self.main.weight, self.main.bias)


def _two_blocks_pool(inp, first, second, Bh, x_act=True):
    """_two_blocks followed by the pool tee of the [moving; fixed] batch, fp32 path: the second block's InstanceNorm apply pass
    also writes the pooled tensor of the next level (ops.instnorm_lrelu_pool_tee_split); returns (pooled, moving, fixed)"""
    raw, st = ops.conv3d_with_stats(inp, first.main.weight, first.main.bias, x_act=x_act)
    raw2, st2 = ops.lazy_instnorm_conv3d(raw, st, second.main.weight, second.main.bias)
    return ops.instnorm_lrelu_pool_tee_split(raw2, st2, Bh)


def _two_blocks(inp, first, second, bf16=False, x_act=True):
    """ConvInsBlock -> ConvInsBlock: the first block's normalised output exists only inside the second conv's kernels
    (ops.lazy_instnorm_conv3d); the second block's InstanceNorm is applied for real (its output has several consumers).
    bf16: the chain's internal tensors are stored in bf16 and the convs run on the bf16 matrix pipe (cfg 5)."""
    if bf16:
        return ops.conv_ins_pair_bf16(inp, first.main.weight, first.main.bias, second.main.weight, second.main.bias)
    # x_act: inp is a normalised block's output or a pooled copy of one (bounded).  Level 1's inp is the ConvBlock 1 -> 4 output --
    # LeakyReLU(conv(image)), as large as the image is -- and goes in with x_act=False: no assumption about its range
    raw, st = ops.conv3d_with_stats(inp, first.main.weight, first.main.bias, x_act=x_act)
    raw2, st2 = ops.lazy_instnorm_conv3d(raw, st, second.main.weight, second.main.bias)
    return ops._InstNormLReLU.apply(raw2, 1e-5, st2)


class _AvgPool(nn.Module):
    def forward(self, x):
        return ops.avgpool2(x)


class Encoder(nn.Module):
    """five-level conv pyramid (reference models.py:181-228); channels-last in/out."""

    def __init__(self, in_channel=1, first_out_channel=4, bf16=False):
        super().__init__()
        self.bf16 = bf16
        c = first_out_channel
        self.conv0 = nn.Sequential(ConvBlock(in_channel, c), ConvInsBlock(c, 2 * c), ConvInsBlock(2 * c, 2 * c))
        self.conv1 = nn.Sequential(_AvgPool(), ConvInsBlock(2 * c, 4 * c), ConvInsBlock(4 * c, 4 * c))
        self.conv2 = nn.Sequential(_AvgPool(), ConvInsBlock(4 * c, 8 * c), ConvInsBlock(8 * c, 8 * c))
        self.conv3 = nn.Sequential(_AvgPool(), ConvInsBlock(8 * c, 16 * c), ConvInsBlock(16 * c, 16 * c))
        self.conv4 = nn.Sequential(_AvgPool(), ConvInsBlock(16 * c, 32 * c), ConvInsBlock(32 * c, 32 * c))

    def forward(self, x):
        # each level's output goes to the next level (pooled) AND to the caller: pool_tee fuses the two gradient paths
        outs = []
        cur = _two_blocks(self.conv0[0](x), self.conv0[1], self.conv0[2], self.bf16, x_act=False)
        for blk in (self.conv1, self.conv2, self.conv3, self.conv4):
            pooled, keep = ops.pool_tee(cur)
            outs.append(keep)
            cur = _two_blocks(pooled, blk[1], blk[2], self.bf16)    # blk[0] is the AvgPool3d(2) the tee already applied
        outs.append(cur)
        return tuple(outs)

    stage_cut = False       # engine.Trainer (overlapped all-reduce): cut the autograd graph in front of level 3
    features16 = False      # ModeT(act_dtype=bfloat16) with the fused level nodes: level features 1-4 stored as bf16 (fp32 handles)

    def _cut(self, t):
        """graph cut for a staged backward: the consumer sees a fresh leaf sharing t's storage (no copy); the stage that owns
        the producer later backpropagates ``leaf.grad`` into ``t`` (self.cut = (t, leaf))"""
        leaf = t.detach().requires_grad_(True)
        self.cut = (t, leaf)
        return leaf

    def forward_pair(self, x, B, handles=False):
        """x = [moving; fixed] as one batch of 2B (InstanceNorm is per sample, so this is exact); returns the per-level
        features of each half: ([M1..M5], [F1..F5]).  ``handles`` (ModeT.forward with the fused bf16 level nodes only): levels
        1-4 come back as fp32 HANDLES carrying bf16 data (``.data16``) that only ops.level_attention_bf16 can read; every other
        caller gets real fp32 features whatever ``features16`` says."""
        features16 = self.features16 and handles
        Ms, Fs = [], []
        pooled_in = []
        if not self.bf16:
            # fp32: the last InstanceNorm of a level writes the level's features and their pooled copy in one pass
            pooled, m, f = _two_blocks_pool(self.conv0[0](x), self.conv0[1], self.conv0[2], B, x_act=False)
            Ms.append(m)
            Fs.append(f)
            pooled_in.append(pooled)
            for blk in (self.conv1, self.conv2, self.conv3):
                if blk is self.conv2 and self.stage_cut:
                    pooled = self._cut(pooled)
                pooled, m, f = _two_blocks_pool(pooled, blk[1], blk[2], B)
                Ms.append(m)
                Fs.append(f)
                pooled_in.append(pooled)
            cur = _two_blocks(pooled, self.conv4[1], self.conv4[2], False)
        else:
            # bf16 chain: the level's last InstanceNorm writes fp32 features; its BACKWARD forms the pool gradient + the two
            # halves' gradients on the fly (ops.conv_ins_pair_bf16_pool_split)
            inp, blocks = self.conv0[0](x), (self.conv0, self.conv1, self.conv2, self.conv3)
            for lvl, blk in enumerate(blocks):
                if blk is self.conv2 and self.stage_cut:
                    inp = self._cut(inp)
                pooled, m, f = ops.conv_ins_pair_bf16_pool_split(inp, blk[1].main.weight, blk[1].main.bias, blk[2].main.weight,
                                                                 blk[2].main.bias, B, features16=features16)
                Ms.append(m)
                Fs.append(f)
                pooled_in.append(pooled)
                inp = pooled
            cur = _two_blocks(inp, self.conv4[1], self.conv4[2], self.bf16)
        m, f = _SplitBatch.apply(cur, B)
        Ms.append(m)
        Fs.append(f)
        return Ms, Fs


class _LinearParams(nn.Module):
    def __init__(self, cin, dim):
        super().__init__()
        # reference init: weight ~ N(0, 1e-5), bias 0 (models.py:235-236)
        self.weight = nn.Parameter(torch.randn(dim, cin) * 1e-5)
        self.bias = nn.Parameter(torch.zeros(dim))


class _LayerNormParams(nn.Module):
    def __init__(self, dim):
        super().__init__()
        self.weight = nn.Parameter(torch.ones(dim))
        self.bias = nn.Parameter(torch.zeros(dim))


class ProjectionLayer(nn.Module):
    """Linear + LayerNorm (reference models.py:230-241); channels-last (B,D,H,W,C) -> (B,D,H,W,dim)."""

    def __init__(self, in_channels, dim=6):
        super().__init__()
        self.norm = _LayerNormParams(dim)
        self.proj = _LinearParams(in_channels, dim)

    def forward(self, feat):
        return ops.proj_ln(feat, self.proj.weight, self.proj.bias, self.norm.weight, self.norm.bias, 1e-5)

    def forward_pair(self, fixed_feat, moving_feat):
        """(q, k) = (self(fixed_feat), self(moving_feat)): one backward reduction for the shared parameters"""
        return ops.proj_ln_pair(fixed_feat, moving_feat, self.proj.weight, self.proj.bias, self.norm.weight, self.norm.bias,
                                1e-5)


class CWM(nn.Module):
    """competitive weighting module (reference models.py:243-275); channels-last."""

    def __init__(self, in_channels, channels):
        super().__init__()
        self.num_fields = in_channels // 3
        # index 3 of the reference Sequential is nn.Softmax (no parameters)
        self.conv = nn.Sequential(ConvInsBlock(in_channels, channels), ConvInsBlock(channels, channels),
                                  _Conv3dParams(channels, self.num_fields))

    def forward(self, x):
        x = ops.upsample2(x, 1.0)
        # ConvIns -> ConvIns -> Conv: both normalised intermediates live only inside the consuming conv's kernels
        raw0, st0 = ops.conv3d_with_stats(x, self.conv[0].main.weight, self.conv[0].main.bias, x_act=True)   # (x: flow fields, |x| <= 1)
        raw1, st1 = ops.lazy_instnorm_conv3d(raw0, st0, self.conv[1].main.weight, self.conv[1].main.bias)
        logits, _ = ops.lazy_instnorm_conv3d(raw1, st1, self.conv[2].weight, self.conv[2].bias, want_stats=False)
        return ops.cwm_tail(x, logits)


class ModeTransformer(nn.Module):
    """3x3x3 neighbourhood attention whose values are the 27 offsets (reference models.py:278-334)."""

    def __init__(self, dim, num_heads, kernel_size=3, qk_scale=None, use_rpb=True, buffer_flavour="grid",
                 fused=True):
        """``fused=False``: the reference ModeT-cu decomposition -- layout prep, the operator ``modetqkrpb_cu`` (logits),
        softmax, ``attn @ v`` (ModeT-cu/models.py:300-316) -- instead of the single fused kernel; same result."""
        super().__init__()
        self.fused = fused
        if kernel_size != 3:
            raise RuntimeError("ModeTransformer does not support kernel size %d" % kernel_size)
        self.num_heads = num_heads
        self.head_dim = dim // num_heads
        self.scale = qk_scale or self.head_dim ** -0.5
        self.kernel_size = kernel_size
        self.use_rpb = use_rpb
        if use_rpb:
            self.rpb = nn.Parameter(torch.zeros(num_heads, 3, 3, 3))
        r = torch.arange(-1, 2)
        grid = torch.stack(torch.meshgrid(r, r, r, indexing="ij"), -1).float()
        self.buffer_flavour = buffer_flavour
        if buffer_flavour == "grid":
            self.register_buffer("grid", grid)                       # ModeT/models.py:293-296
        else:
            self.register_buffer("v", grid.reshape(27, 3))           # ModeT-cu/models.py:294-298

    def _load_from_state_dict(self, state_dict, prefix, *args, **kwargs):
        have, other = ("grid", "v") if self.buffer_flavour == "grid" else ("v", "grid")
        if prefix + other in state_dict and prefix + have not in state_dict:
            t = state_dict.pop(prefix + other)
            state_dict[prefix + have] = t.reshape(3, 3, 3, 3) if have == "grid" else t.reshape(27, 3)
        super()._load_from_state_dict(state_dict, prefix, *args, **kwargs)

    def forward(self, q, k):
        if not self.fused:
            return self.forward_operator(q, k)
        if self.use_rpb:
            rpb = self.rpb
        else:
            rpb = torch.zeros(self.num_heads, 3, 3, 3, dtype=q.dtype, device=q.device)
        return ops.neighbourhood_attention(q, k, rpb, self.num_heads, self.scale)

    def forward_operator(self, q, k):
        """ModeT-cu/models.py:300-316 statement for statement on channels-last q, k (B,D,H,W,C): the operator boundary
        ``modetqkrpb_cu`` (functional.py -> modet_qk_fwd / modet_qk_bwd) sees exactly the tensors the reference hands its
        CUDA extension -- pre-scaled (B,heads,D,H,W,d) queries, zero-padded (B,heads,D+2,H+2,W+2,d) keys -- and the
        softmax / ``@ v`` / layout steps around it are the reference's own ATen calls.  Returns (B,D,H,W,heads*3)."""
        from .functional import modetqkrpb_cu
        B, D, H, W, C = q.shape
        h, d = self.num_heads, C // self.num_heads
        qh = q.reshape(B, D, H, W, h, d).permute(0, 4, 1, 2, 3, 5) * self.scale
        kk = torch.nn.functional.pad(k.permute(0, 4, 1, 2, 3), (1, 1, 1, 1, 1, 1))
        kk = kk.reshape(B, h, d, D + 2, H + 2, W + 2).permute(0, 1, 3, 4, 5, 2)
        attn = modetqkrpb_cu(qh, kk, self.rpb if self.use_rpb else None)
        attn = attn.softmax(dim=-1)
        v = self.v if self.buffer_flavour == "v" else self.grid.reshape(27, 3)
        x = attn @ v                                              # (B,heads,D,H,W,3)
        return x.permute(0, 2, 3, 4, 1, 5).reshape(B, D, H, W, h * 3)      # channel = head*3 + axis (models.py:314)


class CoTr(nn.Module):
    """Coordinate translator of the Im2Grid baseline ("Baseline methods/Im2Grid/models.py":276-322): the same 3x3x3
    neighbourhood attention with ONE head over all C channels, no bias, no scale -- SURVEY.md 8(f) rank 4: it runs on
    the attention kernels' generic head-dimension path (C a multiple of 8, up to 128).

    ``forward(q, k)``: q, k (B,H,W,T,C) channels-last as in the reference -> expected offset (B,3,H,W,T)."""

    def __init__(self, kernel_size=3):
        super().__init__()
        if kernel_size != 3:
            raise RuntimeError("CoTr does not support kernel size %d" % kernel_size)
        self.kernel_size = kernel_size
        r = torch.arange(-1, 2)
        self.register_buffer("grid", torch.stack(torch.meshgrid(r, r, r, indexing="ij"), -1).float())

    def forward(self, q, k):
        rpb = torch.zeros(1, 3, 3, 3, dtype=q.dtype, device=q.device)
        return ops.to_ncdhw(ops.neighbourhood_attention(q.contiguous(), k.contiguous(), rpb, 1, 1.0))


class Correlation3D(nn.Module):
    """27-displacement local correlation of the PR++ baseline ("Baseline methods/PR++/models.py":205-232; the only
    configuration used there: kernel_size 3, d = 3, sw = 1, sf = 2) -- SURVEY.md 8(f) rank 4.
    ``forward(mov, fix)``: NCDHW features (B,C,H,W,T) -> (B,27,H,W,T); C must be a multiple of 4."""

    def __init__(self, in_channel, kernel_size=3, d=3, sw=1, sf=2):
        super().__init__()
        if (kernel_size, d, sw, sf) != (3, 3, 1, 2):
            raise RuntimeError("Correlation3D supports kernel_size=3, d=3, sw=1, sf=2 only")
        self.in_channel = in_channel

    def forward(self, mov, fix):
        return ops.correlation3d(ops.to_channels_last(mov.contiguous()), ops.to_channels_last(fix.contiguous()))


class ModeT(nn.Module):
    """reference ModeT/models.py:338-412 (scale=None -> head_dim**-0.5)."""

    _flavour = "grid"
    stage_cuts = False      # set by engine.Trainer around a forward whose backward runs in stages

    def __init__(self, inshape=(160, 192, 160), in_channel=1, channels=4, head_dim=6, num_heads=[8, 4, 2, 1, 1],
                 scale=None, legacy_grid_buffers=False, act_dtype=torch.float32, fused_attention=True):
        """``act_dtype=torch.bfloat16`` (not in the reference, BASELINE.json configs[4]): the encoder's ConvInsBlock chains
        store their activations in bf16 and run on the bf16 matrix pipe with fp32 accumulation; parameters, statistics,
        level features, flows, losses and the optimizer stay fp32, so checkpoints are unchanged; each level's warped moving
        features and its q / k projections -- tensors only the level's own matching step reads -- are stored as bf16 too
        (one autograd node per level, ops.level_attention_bf16).
        ``fused_attention=False``: every ModeTransformer runs the reference ModeT-cu decomposition through the operator
        boundary ``modetqkrpb_cu`` (see ModeTransformer.forward_operator); every level dim must then be >= 3, as the
        reference's CHECK_3DFEATMAP demands (utils.h:10)."""
        super().__init__()
        self.fused_attention = fused_attention
        if act_dtype not in (torch.float32, torch.bfloat16):
            raise RuntimeError("ModeT: act_dtype must be torch.float32 or torch.bfloat16")
        self.act_dtype = act_dtype
        # bf16 storage of every level's warped features, q and k (ops.level_attention_bf16): head_dim 6 and the model's channel
        # counts only (the kernels the fused node runs), with the fused attention
        self.level_bf16 = (act_dtype == torch.bfloat16 and fused_attention and head_dim == 6 and channels == 4)
        if self.level_bf16:
            # the fused level node's backward is the PAIRED projection backward, which exists for the (C_in, dim) pairs of
            # the grouped kernels only (the default num_heads gives them at every level); another head layout keeps the bf16
            # conv chains and runs the levels on fp32 features (it used to raise in backward)
            L = ops._L()
            pairs = [(2 * channels * 2 ** i, head_dim * num_heads[4 - i]) for i in range(5)]
            self.level_bf16 = all(L.modet_proj_ln_bwd_pair_ws_bytes(4096, cin, dim) > 0 for cin, dim in pairs)
        # like the reference's constructor (models.py:338-375) any inshape is accepted here; shapes that four 2x poolings
        # do not divide fail in forward(), where the reference fails too (its x2-upsampled flow no longer matches the
        # next level's grid: "The size of tensor a must match the size of tensor b")
        if len(tuple(inshape)) != 3 or any(int(v) < 1 for v in inshape):
            raise RuntimeError("ModeT: inshape = (D, H, W)")
        self.channels = channels
        self.step = 7
        self.inshape = tuple(inshape)
        c = channels
        fl = self._flavour
        mk = dict(qk_scale=scale, buffer_flavour=fl, fused=fused_attention)
        self.encoder = Encoder(in_channel=in_channel, first_out_channel=c, bf16=act_dtype == torch.bfloat16)
        # the level features of levels 1-4 as bf16 too: only the fused level nodes read them (as ``.data16`` of fp32 handles)
        self.encoder.features16 = self.level_bf16
        self.projblock1 = ProjectionLayer(2 * c, dim=head_dim * num_heads[4])
        self.mdt1 = ModeTransformer(head_dim * num_heads[4], num_heads[4], **mk)
        self.projblock2 = ProjectionLayer(4 * c, dim=head_dim * num_heads[3])
        self.mdt2 = ModeTransformer(head_dim * num_heads[3], num_heads[3], **mk)
        self.projblock3 = ProjectionLayer(8 * c, dim=head_dim * num_heads[2])
        self.mdt3 = ModeTransformer(head_dim * num_heads[2], num_heads[2], **mk)
        self.cwm3 = CWM(3 * num_heads[2], 3 * num_heads[2] * 2)
        self.projblock4 = ProjectionLayer(16 * c, dim=head_dim * num_heads[1])
        self.mdt4 = ModeTransformer(head_dim * num_heads[1], num_heads[1], **mk)
        self.cwm4 = CWM(3 * num_heads[1], 3 * num_heads[1] * 2)
        self.projblock5 = ProjectionLayer(32 * c, dim=head_dim * num_heads[0])
        self.mdt5 = ModeTransformer(head_dim * num_heads[0], num_heads[0], **mk)
        self.cwm5 = CWM(3 * num_heads[0], 3 * num_heads[0] * 2)
        self.transformer = nn.ModuleList(
            [SpatialTransformer([s // 2 ** i for s in inshape], legacy_grid_buffers=legacy_grid_buffers)
             for i in range(4)])

    def forward(self, moving, fixed):
        """(y_moved (B,1,D,H,W), flow (B,3,D,H,W)) -- reference models.py:380-412"""
        y_moved, flow = self.forward_cl(moving, fixed)
        return ops.to_ncdhw(y_moved), ops.to_ncdhw(flow)

    def forward_cl(self, moving, fixed):
        """``forward`` with both results in the layout the kernels wrote them: y_moved (B,D,H,W,1), flow (B,D,H,W,3).  The
        trainer takes these (its Grad3d kernel reads the channels-last flow and writes the gradient in place of the planar
        round trip); ``forward`` adds the reference's (B,C,D,H,W) views / copies."""
        if moving.shape != fixed.shape or moving.dim() != 5:
            raise RuntimeError("ModeT.forward expects two (B,C,D,H,W) volumes of the same shape")
        if any(int(v) % 16 != 0 for v in moving.shape[2:]):
            raise RuntimeError(f"ModeT.forward: every spatial dimension must be a multiple of 16 (four 2x poolings, then x2 "
                               f"upsampling back: the level sizes would not match), got {tuple(moving.shape[2:])}")
        B = moving.shape[0]
        mov_cl = ops.to_channels_last(moving.contiguous())
        fix_cl = ops.to_channels_last(fixed.contiguous())
        # shared encoder on both images as one batch (InstanceNorm is per sample, so this is exact)
        with ops.trace_range("encoder"):
            M, Fx = self.encoder.forward_pair(ops.cat_batch(mov_cl, fix_cl), B, handles=self.level_bf16)
        if self.stage_cuts:
            # staged backward (engine.Trainer, overlapped all-reduce): the heads consume fresh LEAVES that share the features'
            # storage, so the heads' graph and the encoder's are disconnected and each stage's autograd run touches its own
            # nodes only (a cut at a non-leaf does not do that: autograd marks everything that can reach a node with a
            # captured output as needed, and the first stage would run -- and free -- half of the encoder's backward)
            self.cut_features = (M, Fx)
            M = [ops.feature_handle_like(m.detach().requires_grad_(True), m) for m in M]
            Fx = [ops.feature_handle_like(f.detach().requires_grad_(True), f) for f in Fx]
            self.cut_leaves = (M, Fx)
        ST = self.transformer

        def match(lvl, proj, mdt, flow):
            """the level's matching step: attention(proj(F), proj(warp(M, flow))) -> (expected offset field, flow).  The
            RETURNED flow is what the caller goes on with: the flow has a second consumer (the next composition), and the
            feature warp's node adds that consumer's gradient inside its own backward kernel (ops.warp_tee).  bf16 storage
            mode (act_dtype=bfloat16, fused attention): one node whose warped features, q and k are bf16 in HBM"""
            if self.level_bf16 and mdt.fused and mdt.use_rpb:
                res = ops.level_attention_bf16(Fx[lvl], M[lvl], flow, proj.proj.weight, proj.proj.bias, proj.norm.weight,
                                               proj.norm.bias, mdt.rpb, mdt.num_heads, mdt.scale, tee=flow is not None)
                return res if flow is not None else (res, None)
            if getattr(Fx[lvl], "data16", None) is not None:
                raise RuntimeError("ModeT: bf16 level features are only readable by the fused level nodes")
            if flow is None:
                Mw = M[lvl]
            else:
                Mw, flow = ops.warp_tee(M[lvl], flow)
            q, k = proj.forward_pair(Fx[lvl], Mw)
            return mdt(q, k), flow

        with ops.trace_range("level5"):
            flow = self.cwm5(match(4, self.projblock5, self.mdt5, None)[0])

        with ops.trace_range("level4"):
            field, flow = match(3, self.projblock4, self.mdt4, flow)
            w = self.cwm4(field)
            flow = ST[2].forward_cl(ops.upsample2(flow, 2.0), w, add_flow=True)

        with ops.trace_range("level3"):
            field, flow = match(2, self.projblock3, self.mdt3, flow)
            w = self.cwm3(field)
            flow = ST[1].forward_cl(ops.upsample2(flow, 2.0), w, add_flow=True)

        with ops.trace_range("level2"):
            w, flow = match(1, self.projblock2, self.mdt2, flow)
            # w comes straight from the attention (expected offset in [-1,1]^3): bounded-flow backward, no atomics
            flow = ops.upsample2(ST[1].forward_cl(flow, w, add_flow=True, flow_bound=1), 2.0)

        with ops.trace_range("level1"):
            w, flow = match(0, self.projblock1, self.mdt1, flow)
            flow = ST[0].forward_cl(flow, w, add_flow=True, flow_bound=1)
            y_moved, flow = ops.warp_tee(mov_cl, flow)      # (the flow's other consumer: the caller, Grad3d in training)
        return y_moved, flow


class ModeT_cu(ModeT):
    """reference ModeT-cu/models.py:319-393: same network, ``scale=1`` default, ``mdtN.v`` buffers."""

    _flavour = "v"

    def __init__(self, inshape=(160, 192, 160), in_channel=1, channels=4, head_dim=6, num_heads=[8, 4, 2, 1, 1],
                 scale=1, legacy_grid_buffers=False, act_dtype=torch.float32, fused_attention=True):
        super().__init__(inshape, in_channel, channels, head_dim, num_heads, scale, legacy_grid_buffers, act_dtype,
                         fused_attention)


def load_numpy_weights(model: nn.Module, weights) -> None:
    """copy a name -> ndarray dict (smilecode_amd.synth.make_weights) into the model's parameters"""
    params = dict(model.named_parameters())
    with torch.no_grad():
        for name, arr in weights.items():
            params[name].copy_(torch.from_numpy(arr).to(params[name].device))
