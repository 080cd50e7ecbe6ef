"""helpers shared by the parity tests"""
import json
import os

import numpy as np
import torch

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def gold(name):
    return np.load(os.path.join(GOLD, name))


def dev():
    return torch.device("cuda:0")


def cl(a):
    """numpy NCDHW -> cuda float32 channels-last contiguous"""
    t = torch.from_numpy(np.ascontiguousarray(a)).float()
    return t.permute(0, 2, 3, 4, 1).contiguous().to(dev())


def cu(a):
    return torch.from_numpy(np.ascontiguousarray(a)).float().contiguous().to(dev())


def ncdhw(t):
    """cuda channels-last -> numpy NCDHW float64"""
    return t.detach().permute(0, 4, 1, 2, 3).double().cpu().numpy()


def np64(t):
    return t.detach().double().cpu().numpy()


def assert_close(got, want, atol=2e-5, rtol=2e-5, what=""):
    got = np.asarray(got, dtype=np.float64)
    want = np.asarray(want, dtype=np.float64)
    assert got.shape == want.shape, f"{what}: shape {got.shape} vs {want.shape}"
    err = np.abs(got - want)
    tol = atol + rtol * np.abs(want)
    bad = err > tol
    if bad.any():
        i = np.unravel_index(np.argmax(err - tol), err.shape)
        raise AssertionError(f"{what}: {bad.sum()}/{bad.size} elements off; worst at {i}: got {got[i]!r} want {want[i]!r} "
                             f"(|err| {err[i]:.3e}, max|want| {np.abs(want).max():.3e})")
    return float(err.max())


_REPORT_PATH = os.path.join(os.path.dirname(GOLD.rstrip("/")), "..", "gpurun_out", "parity_report.json")


def note(key, val):
    """record a measured parity number in gpurun_out/parity_report.json; the file is MERGED (keys of other test
    modules / earlier subsets of the same call survive), so whichever subset ran last does not erase the rest"""
    path = os.path.normpath(_REPORT_PATH)
    os.makedirs(os.path.dirname(path), exist_ok=True)
    rep = {}
    if os.path.exists(path):
        try:
            rep = json.load(open(path))
        except (OSError, ValueError):
            rep = {}
    rep[key] = float(val)
    with open(path, "w") as f:
        json.dump(rep, f, indent=1, sort_keys=True)
