"""Input pipeline of the LPBA experiments (reference ModeT/data/datasets.py:8-93, data/trans.py:27-55),
re-stated: `.pkl` files holding `(image float32 (D,H,W), label uint16 (D,H,W))`, every ordered pair of
subjects is one sample ("S2S").  Host-side I/O only -- no arithmetic of the hot path lives here."""
from __future__ import annotations

import glob
import pickle

import numpy as np
import torch
from torch.utils.data import Dataset

# LPBA40 label ids -> 0..54 (reference data/trans.py:30-32)
LPBA_LABEL_IDS = (0, 21, 22, 23, 24, 25, 26, 27, 28, 29, 30, 31, 32, 33, 34, 41, 42, 43, 44, 45, 46, 47, 48, 49, 50, 61,
                  62, 63, 64, 65, 66, 67, 68, 81, 82, 83, 84, 85, 86, 87, 88, 89, 90, 91, 92, 101, 102, 121, 122, 161,
                  162, 163, 164, 165, 166)


def pkload(fname):
    with open(fname, "rb") as f:
        return pickle.load(f)


def pair_indices(index: int, n: int):
    """sample index -> (moving subject, fixed subject) over all n*(n-1) ordered pairs (datasets.py:24-26)"""
    x = index // (n - 1)
    s = index % (n - 1)
    return x, (s + 1 if s >= x else s)


def seg_norm(label: np.ndarray) -> np.ndarray:
    """remap LPBA ids to 0..54; ids outside the table become 0 (trans.py:34-39), as one table lookup"""
    lut = np.zeros(int(max(LPBA_LABEL_IDS)) + 1, dtype=np.int16)
    for i, v in enumerate(LPBA_LABEL_IDS):
        lut[v] = i
    lab = np.asarray(label).astype(np.int64)
    out = np.zeros(lab.shape, dtype=np.int16)
    ok = (lab >= 0) & (lab < lut.size)
    out[ok] = lut[lab[ok]]
    return out


class LPBABrainDatasetS2S(Dataset):
    """training pairs: returns (moving, fixed) as (1,D,H,W) float32 tensors (datasets.py:12-55)"""

    def __init__(self, data_path, transforms=None):
        self.paths = sorted(data_path) if not isinstance(data_path, str) else sorted(glob.glob(data_path))
        self.transforms = transforms

    def __len__(self):
        return len(self.paths) * (len(self.paths) - 1)

    def __getitem__(self, index):
        xi, yi = pair_indices(index, len(self.paths))
        x, _ = pkload(self.paths[xi])
        y, _ = pkload(self.paths[yi])
        x, y = np.asarray(x, np.float32)[None], np.asarray(y, np.float32)[None]
        return torch.from_numpy(np.ascontiguousarray(x)), torch.from_numpy(np.ascontiguousarray(y))


class LPBABrainInferDatasetS2S(LPBABrainDatasetS2S):
    """validation pairs: (moving, fixed, moving labels, fixed labels), labels remapped to 0..54 int16 (datasets.py:58-93)"""

    def __getitem__(self, index):
        xi, yi = pair_indices(index, len(self.paths))
        x, xs = pkload(self.paths[xi])
        y, ys = pkload(self.paths[yi])
        x, y = np.asarray(x, np.float32)[None], np.asarray(y, np.float32)[None]
        xs, ys = seg_norm(xs)[None], seg_norm(ys)[None]
        return tuple(torch.from_numpy(np.ascontiguousarray(a)) for a in (x, y, xs, ys))


class SyntheticPairs(Dataset):
    """stand-in for the LPBA files (absent from the reference repo): seeded blob volumes + 54-label maps of
    smilecode_amd.synth, `n` subjects -> n*(n-1) ordered pairs, same tuple layout as the datasets above"""

    def __init__(self, shape, n=4, seed=24, with_labels=False):
        from . import synth
        self.vols = [synth.make_volume(shape, seed + i)[None] for i in range(n)]
        self.labs = [synth.make_labels(shape, seed + i)[None] for i in range(n)] if with_labels else None

    def __len__(self):
        return len(self.vols) * (len(self.vols) - 1)

    def __getitem__(self, index):
        xi, yi = pair_indices(index, len(self.vols))
        out = [torch.from_numpy(self.vols[xi]), torch.from_numpy(self.vols[yi])]
        if self.labs is not None:
            out += [torch.from_numpy(self.labs[xi]), torch.from_numpy(self.labs[yi])]
        return tuple(out)


class DeviceVolumeCache:
    """All subjects of a pair dataset resident in HBM, pairs indexed there.

    The reference feeds its loop from ``DataLoader(num_workers=4, pin_memory=True)`` (train.py:98-99): two un-pickles and a
    39 MB H2D copy per iteration.  At ~12 ms per train step that loader, not the GPU, would set the pace -- and the
    40 LPBA volumes are 0.8 GB, 0.3 % of one MI355X's HBM.  So: read every subject ONCE (a thread pool un-pickles while the
    previous volume's pinned staging buffer drains over PCIe on a side stream), keep ``(N,1,D,H,W)`` fp32 volumes (and
    int16 label maps, already ``seg_norm``-ed) on the device, and hand out views: a training step then starts with zero
    host work and zero copies.

    ``source``: a ``LPBABrainDatasetS2S`` / ``LPBABrainInferDatasetS2S`` (``.paths``) or a ``SyntheticPairs`` (``.vols``)."""

    def __init__(self, source, device=None, with_labels=False, workers=4):
        from concurrent.futures import ThreadPoolExecutor
        dev = torch.device("cuda", torch.cuda.current_device()) if device is None else torch.device(device)
        self.with_labels = with_labels
        if hasattr(source, "paths"):
            n = len(source.paths)

            def read(i):
                img, lab = pkload(source.paths[i])
                return np.asarray(img, np.float32), (seg_norm(lab) if with_labels else None)
        else:
            n = len(source.vols)

            def read(i):
                return source.vols[i][0], (source.labs[i][0] if with_labels else None)
        if n < 2:
            raise RuntimeError("DeviceVolumeCache: a pair dataset needs at least two subjects")
        self.n = n
        self.vols = self.labs = None
        use_cuda = dev.type == "cuda"
        side = torch.cuda.Stream(dev) if use_cuda else None
        stage, events = [None, None], [None, None]
        with ThreadPoolExecutor(max_workers=max(1, workers)) as ex:
            for i, (img, lab) in enumerate(ex.map(read, range(n))):
                if self.vols is None:
                    self.vols = torch.empty((n, 1) + img.shape, dtype=torch.float32, device=dev)
                    if with_labels:
                        self.labs = torch.empty((n, 1) + img.shape, dtype=torch.int16, device=dev)
                if not use_cuda:
                    self.vols[i, 0].copy_(torch.from_numpy(np.ascontiguousarray(img)))
                    if with_labels:
                        self.labs[i, 0].copy_(torch.from_numpy(np.ascontiguousarray(lab)))
                    continue
                s = i & 1                               # two pinned staging slots: fill one while the other drains
                if events[s] is not None:
                    events[s].synchronize()
                if stage[s] is None:
                    stage[s] = (torch.empty(img.shape, dtype=torch.float32).pin_memory(),
                                torch.empty(img.shape, dtype=torch.int16).pin_memory() if with_labels else None)
                stage[s][0].copy_(torch.from_numpy(np.ascontiguousarray(img)))
                if with_labels:
                    stage[s][1].copy_(torch.from_numpy(np.ascontiguousarray(lab)))
                with torch.cuda.stream(side):
                    self.vols[i, 0].copy_(stage[s][0], non_blocking=True)
                    if with_labels:
                        self.labs[i, 0].copy_(stage[s][1], non_blocking=True)
                    events[s] = torch.cuda.Event()
                    events[s].record(side)
        if use_cuda:
            side.synchronize()

    def __len__(self):
        return self.n * (self.n - 1)

    def pair(self, index):
        """(moving, fixed) as (1,1,D,H,W) device views [+ (moving labels, fixed labels) int16] of sample ``index``"""
        xi, yi = pair_indices(index, self.n)
        out = (self.vols[xi:xi + 1], self.vols[yi:yi + 1])
        if self.with_labels:
            out += (self.labs[xi:xi + 1], self.labs[yi:yi + 1])
        return out
