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
