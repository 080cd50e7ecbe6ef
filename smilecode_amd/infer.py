"""Inference / evaluation script with the reference's flow and print lines (ModeT/infer.py:49-101):
load the best checkpoint, register every validation pair, nearest-warp the labels, Dice over the 54 VOIs and the
fraction of voxels with non-positive Jacobian determinant.

    python -m smilecode_amd.infer --val-dir /LPBA_path/Val/ --model-dir experiments/<run>/
    python -m smilecode_amd.infer --synthetic 3 --img-size 64,64,64            # random weights, no data needed
"""
from __future__ import annotations

import argparse
import glob
import os

import torch

from . import data, utils
from .models import ModeT
from .train import _natkey


def main(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument("--val-dir", default="/LPBA_path/Val/")
    ap.add_argument("--model-dir", default="")
    ap.add_argument("--model-idx", type=int, default=-1)
    ap.add_argument("--synthetic", type=int, default=0)
    ap.add_argument("--img-size", default="160,192,160")
    args = ap.parse_args(argv)
    img_size = tuple(int(s) for s in args.img_size.split(","))
    model = ModeT(img_size, head_dim=6, num_heads=[8, 4, 2, 1, 1], scale=1)
    if args.model_dir:
        files = sorted(os.listdir(args.model_dir), key=_natkey)
        # weights_only=False: checkpoints written by the reference's train.py carry numpy scalars (see train.py here)
        best = torch.load(os.path.join(args.model_dir, files[args.model_idx]), map_location="cpu",
                          weights_only=False)["state_dict"]
        print("Best model: {}".format(files[args.model_idx]))
        model.load_state_dict(best)
    model.cuda().eval()
    if args.synthetic:
        test_set = data.SyntheticPairs(img_size, args.synthetic, 124, with_labels=True)
    else:
        test_set = data.LPBABrainInferDatasetS2S(glob.glob(args.val_dir + "*.pkl"))
    cache = data.DeviceVolumeCache(test_set, with_labels=True)      # every subject resident in HBM, pairs indexed there
    eval_dsc_def, eval_dsc_raw, eval_det = utils.AverageMeter(), utils.AverageMeter(), utils.AverageMeter()
    with torch.no_grad():
        for i in range(len(cache)):
            x, y, x_seg, y_seg = cache.pair(i)
            _, flow = model(x, y)
            _, dsc_trans = utils.warp_labels_and_dice(x_seg, flow, y_seg)
            dsc_raw = float(utils.dice_val_VOI(x_seg, y_seg))
            # infer.py:89-90 without the 59 MB D2H + numpy pass: one integer per pair comes back from the GPU
            eval_det.update(utils.jacobian_nonpositive_fraction(flow)[0], x.size(0))
            print("Trans dsc: {:.4f}, Raw dsc: {:.4f}".format(dsc_trans, dsc_raw))
            eval_dsc_def.update(dsc_trans, x.size(0))
            eval_dsc_raw.update(dsc_raw, x.size(0))
    print("Deformed DSC: {:.3f} +- {:.3f}, Affine DSC: {:.3f} +- {:.3f}".format(eval_dsc_def.avg, eval_dsc_def.std,
                                                                                eval_dsc_raw.avg, eval_dsc_raw.std))
    print("deformed det: {}, std: {}".format(eval_det.avg, eval_det.std))
    return eval_dsc_def.avg


if __name__ == "__main__":
    main()
