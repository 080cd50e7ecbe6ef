"""Seeded synthetic volumes, labels and weights (numpy only, no torch RNG).

Everything the bench, the parity tests and the golden-vector generator feed the
hot path comes from here, so the three see bit-identical inputs on any machine
(numpy's PCG64 stream is platform independent).  Recipe follows SURVEY.md §8(d):

* volumes: sum of anisotropic Gaussian blobs + 0.05*U(0,1) noise, times an
  ellipsoidal "skull-stripped" mask (exact zeros outside, as the LPBA min-max
  data has: reference makePklDataset.py:19-20,:76), values in [0,1];
* labels: 54 regions = 3x6x3 block partition of the mask, int16, 0 = background
  (reference utils.py:86-106 scores labels 1..54);
* weights: the parameter list of reference ModeT/models.py:338-375 with
  non-degenerate projection / rpb values (the reference's own init,
  models.py:235, makes the attention uniform and the flow ~0).
"""
from __future__ import annotations

import zlib
from collections import OrderedDict

import numpy as np

NUM_HEADS_DEFAULT = (8, 4, 2, 1, 1)


def param_spec(channels: int = 4, head_dim: int = 6, num_heads=NUM_HEADS_DEFAULT,
               in_channel: int = 1) -> "OrderedDict[str, tuple]":
    """Ordered parameter name -> shape map, identical to the reference
    ``ModeT(...).named_parameters()`` (ModeT/models.py:338-375; dump in SURVEY.md §5)."""
    c = channels
    spec: "OrderedDict[str, tuple]" = OrderedDict()

    def conv(name, cin, cout):
        spec[name + ".weight"] = (cout, cin, 3, 3, 3)
        spec[name + ".bias"] = (cout,)

    conv("encoder.conv0.0.main", in_channel, c)
    conv("encoder.conv0.1.main", c, 2 * c)
    conv("encoder.conv0.2.main", 2 * c, 2 * c)
    for lvl in range(1, 5):
        cin, cout = (2 ** lvl) * c, (2 ** (lvl + 1)) * c
        conv(f"encoder.conv{lvl}.1.main", cin, cout)
        conv(f"encoder.conv{lvl}.2.main", cout, cout)
    for lvl in range(1, 6):
        heads = num_heads[5 - lvl]
        dim = head_dim * heads
        cin = (2 ** lvl) * c
        spec[f"projblock{lvl}.norm.weight"] = (dim,)
        spec[f"projblock{lvl}.norm.bias"] = (dim,)
        spec[f"projblock{lvl}.proj.weight"] = (dim, cin)
        spec[f"projblock{lvl}.proj.bias"] = (dim,)
        spec[f"mdt{lvl}.rpb"] = (heads, 3, 3, 3)
        if lvl >= 3:
            cc = 3 * heads
            conv(f"cwm{lvl}.conv.0.main", cc, 2 * cc)
            conv(f"cwm{lvl}.conv.1.main", 2 * cc, 2 * cc)
            conv(f"cwm{lvl}.conv.2", 2 * cc, heads)
    return spec


def _rng(seed: int, name: str) -> np.random.Generator:
    return np.random.default_rng([int(seed), zlib.crc32(name.encode())])


def make_weights(seed: int = 24, channels: int = 4, head_dim: int = 6,
                 num_heads=NUM_HEADS_DEFAULT, in_channel: int = 1,
                 dtype=np.float32) -> "OrderedDict[str, np.ndarray]":
    """Non-degenerate weights: conv ~ U(+-1/sqrt(fan_in)) (torch's default bound),
    proj.weight ~ N(0, 0.3), proj.bias ~ N(0, 0.1), LayerNorm weight 1+N(0,0.1),
    bias N(0,0.1), rpb ~ N(0, 0.5).  Each tensor has its own stream keyed by name."""
    out: "OrderedDict[str, np.ndarray]" = OrderedDict()
    for name, shape in param_spec(channels, head_dim, num_heads, in_channel).items():
        g = _rng(seed, name)
        if name.endswith("rpb"):
            v = g.normal(0.0, 0.5, shape)
        elif ".norm.weight" in name:
            v = 1.0 + g.normal(0.0, 0.1, shape)
        elif ".norm.bias" in name:
            v = g.normal(0.0, 0.1, shape)
        elif ".proj.weight" in name:
            v = g.normal(0.0, 0.3, shape)
        elif ".proj.bias" in name:
            v = g.normal(0.0, 0.1, shape)
        elif name.endswith(".weight"):
            fan_in = shape[1] * 27
            b = 1.0 / np.sqrt(fan_in)
            v = g.uniform(-b, b, shape)
        else:  # conv bias: bound from the matching weight's fan-in
            wshape = param_spec(channels, head_dim, num_heads, in_channel)[name[:-4] + "weight"]
            b = 1.0 / np.sqrt(wshape[1] * 27)
            v = g.uniform(-b, b, shape)
        out[name] = np.ascontiguousarray(v, dtype=dtype)
    return out


def _mask(shape) -> np.ndarray:
    D, H, W = shape
    z, y, x = np.meshgrid(np.linspace(-1, 1, D), np.linspace(-1, 1, H), np.linspace(-1, 1, W),
                          indexing="ij", sparse=True)
    return ((z / 0.92) ** 2 + (y / 0.88) ** 2 + (x / 0.9) ** 2) <= 1.0


def make_volume(shape, seed: int, dtype=np.float32) -> np.ndarray:
    """One (D,H,W) volume in [0,1] with exact zeros outside an ellipsoid."""
    D, H, W = shape
    g = np.random.default_rng([int(seed), 7])
    z, y, x = np.meshgrid(np.linspace(-1, 1, D), np.linspace(-1, 1, H), np.linspace(-1, 1, W),
                          indexing="ij", sparse=True)
    vol = np.zeros(shape, np.float64)
    for _ in range(6):
        c = g.uniform(-0.5, 0.5, 3)
        s = g.uniform(0.15, 0.6, 3)
        a = g.uniform(0.3, 1.0)
        vol = vol + a * np.exp(-(((z - c[0]) / s[0]) ** 2 + ((y - c[1]) / s[1]) ** 2
                                 + ((x - c[2]) / s[2]) ** 2))
    vol = vol / vol.max()
    vol = 0.95 * vol + 0.05 * g.random(shape)
    vol = vol * _mask(shape)
    return np.ascontiguousarray(vol, dtype=dtype)


def make_pair(shape, seed: int = 24, batch: int = 1, dtype=np.float32):
    """(moving, fixed), each (B,1,D,H,W); moving from ``seed`` (train.py:29 uses 24),
    fixed from ``seed+1``; batch element b uses seed + 2*b."""
    mov = np.stack([make_volume(shape, seed + 2 * b, dtype) for b in range(batch)])[:, None]
    fix = np.stack([make_volume(shape, seed + 2 * b + 1, dtype) for b in range(batch)])[:, None]
    return mov, fix


def make_labels(shape, seed: int = 24) -> np.ndarray:
    """(D,H,W) int16 label map: 54 = 3x6x3 blocks inside the mask, 0 outside.
    ``seed`` jitters the block boundaries so moving/fixed label maps differ."""
    D, H, W = shape
    g = np.random.default_rng([int(seed), 11])

    def edges(n, parts):
        e = np.linspace(0, n, parts + 1)
        e[1:-1] += g.uniform(-0.06, 0.06, parts - 1) * n
        return np.clip(np.searchsorted(e[1:-1], np.arange(n), side="right"), 0, parts - 1)

    bz, by, bx = edges(D, 3), edges(H, 6), edges(W, 3)
    lab = 1 + (bz[:, None, None] * 6 + by[None, :, None]) * 3 + bx[None, None, :]
    lab = lab * _mask(shape)
    return np.ascontiguousarray(lab, dtype=np.int16)


def make_flow(shape, seed: int = 3, amp: float = 3.0, batch: int = 1, dtype=np.float32) -> np.ndarray:
    """Smooth-ish random displacement field (B,3,D,H,W) in voxel units whose border
    samples leave the volume (exercises the zero-padding branch of the warp)."""
    D, H, W = shape
    g = np.random.default_rng([int(seed), 13])
    coarse = g.normal(0.0, amp, (batch, 3, max(D // 4, 2), max(H // 4, 2), max(W // 4, 2)))
    idx = [np.minimum(np.arange(n) * c.shape[0] // n, c.shape[0] - 1)
           for n, c in ((D, np.empty(coarse.shape[2])), (H, np.empty(coarse.shape[3])),
                        (W, np.empty(coarse.shape[4])))]
    f = coarse[:, :, idx[0]][:, :, :, idx[1]][:, :, :, :, idx[2]]
    f = f + g.normal(0.0, 0.25, f.shape)
    return np.ascontiguousarray(f, dtype=dtype)
