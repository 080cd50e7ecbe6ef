"""Golden vectors for the input pipeline, generated from the REAL reference (build container only).

    python tests/golden/make_goldens_data.py

Imports /root/reference/ModeT/data/{datasets,trans}.py read-only (never copied, never shipped) and runs them on four small
synthetic subjects written as the `.pkl` files the reference reads (`(image float32 (D,H,W), label uint16 (D,H,W))`,
data/datasets.py:8-10, makePklDataset.py:8-10):
  * ``datasets.LPBABrainDatasetS2S`` with train.py:92's transforms -> the ordered-pair order and the (moving, fixed) volumes of
    every sample index (datasets.py:23-55);
  * ``datasets.LPBABrainInferDatasetS2S`` with train.py:94-95 / infer.py:68-69's transforms (``trans.Seg_norm`` then
    ``trans.NumpyType((float32, int16))``) -> the four tensors of every sample index (datasets.py:68-90, trans.py:27-39).
Label maps hold every id of the LPBA table plus ids OUTSIDE it (7, 100, 167, 181, 200, 65535): ``Seg_norm`` sends those to 0.
Stand-ins for what this image lacks, neither of which touches the arithmetic: ``matplotlib`` (imported by datasets.py:4, never
used by the classes) is an empty module; ``torchvision.transforms.Compose`` is absent, so the two transforms are applied one
after the other, which is what Compose does; ``collections.Sequence`` (trans.py:19, removed in Python 3.10) is aliased to
``collections.abc.Sequence``.
Writes tests/golden/data_pipeline.npz (data only): the subjects, and per sample index the subject pair and the remapped labels.
"""
import collections
import collections.abc
import os
import pickle
import sys
import tempfile
import types

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
collections.Sequence = collections.abc.Sequence                     # trans.py:19 on Python >= 3.10
sys.modules.setdefault("matplotlib", types.ModuleType("matplotlib"))
sys.modules.setdefault("matplotlib.pyplot", types.ModuleType("matplotlib.pyplot"))
sys.path.insert(0, "/root/reference/ModeT")
from data import datasets as ref_datasets      # noqa: E402  (reference)
from data import trans as ref_trans            # noqa: E402  (reference)


def compose(ts):
    def run(x):
        for t in ts:
            x = t(x)
        return x
    return run


def main():
    rng = np.random.default_rng(2024)
    n, shape = 4, (8, 12, 10)
    table = ref_trans.Seg_norm().seg_table
    ids = np.concatenate([table, np.array([7, 100, 167, 181, 200, 65535])]).astype(np.uint16)
    imgs = rng.random((n,) + shape, dtype=np.float32)
    labs = ids[rng.integers(0, len(ids), size=(n,) + shape)].astype(np.uint16)
    with tempfile.TemporaryDirectory() as d:
        paths = []
        for i in range(n):
            p = os.path.join(d, "S%02d.pkl" % i)
            with open(p, "wb") as f:
                pickle.dump((imgs[i], labs[i]), f)
            paths.append(p)
        train = ref_datasets.LPBABrainDatasetS2S(paths, transforms=compose([ref_trans.NumpyType((np.float32, np.float32))]))
        val = ref_datasets.LPBABrainInferDatasetS2S(paths, transforms=compose([ref_trans.Seg_norm(), ref_trans.NumpyType((np.float32, np.int16))]))
        assert len(train) == len(val) == n * (n - 1)
        pair = np.zeros((len(train), 2), np.int64)
        seg = np.zeros((n,) + shape, np.int16)
        for k in range(len(train)):
            x, y = train[k]
            xv, yv, xs, ys = val[k]
            xi = [i for i in range(n) if np.array_equal(x.numpy()[0], imgs[i])]
            yi = [i for i in range(n) if np.array_equal(y.numpy()[0], imgs[i])]
            assert len(xi) == 1 and len(yi) == 1 and x.dtype.is_floating_point and tuple(x.shape) == (1,) + shape
            assert np.array_equal(xv.numpy(), x.numpy()) and np.array_equal(yv.numpy(), y.numpy())
            assert str(xs.dtype) == "torch.int16" and tuple(xs.shape) == (1,) + shape
            pair[k] = (xi[0], yi[0])
            for i, s in ((xi[0], xs), (yi[0], ys)):
                if k > 0 and seg[i].any():
                    assert np.array_equal(seg[i], s.numpy()[0])           # the remap of a subject does not depend on the pair
                seg[i] = s.numpy()[0]
    out = os.path.join(HERE, "data_pipeline.npz")
    np.savez_compressed(out, imgs=imgs, labs=labs, pair=pair, seg=seg, table=np.asarray(table, np.int64))
    print("wrote", out, "pairs", pair.tolist(), "labels outside the table ->", sorted(set(seg[np.isin(labs, [7, 100, 167, 181, 200, 65535])].ravel().tolist())))


if __name__ == "__main__":
    main()
