"""Training script with the reference's hyper-parameters, loop shape, log lines and checkpoint policy
(ModeT/train.py:42-176), on the MI355X-native path.  One process per GPU:

    python -m smilecode_amd.train --train-dir /LPBA_path/Train/ --val-dir /LPBA_path/Val/
    python -m torch.distributed.run --nproc-per-node 8 --master-addr 127.0.0.1 -m smilecode_amd.train ...
    python -m smilecode_amd.train --synthetic 4 --img-size 64,64,64 --max-epoch 1      # no data needed
"""
from __future__ import annotations

import argparse
import glob
import os
import random
import re
import sys

import numpy as np
import torch

from . import data, utils
from .engine import Trainer, poly_lr
from .models import ModeT
from .parallel import init_from_env, lockstep_pairs_for_rank


def same_seeds(seed):
    random.seed(seed)
    np.random.seed(seed)
    torch.manual_seed(seed)
    if torch.cuda.is_available():
        torch.cuda.manual_seed_all(seed)


class Logger(object):
    """tee stdout into logs/<save_dir>/logfile.log (train.py:30-40)"""

    def __init__(self, save_dir):
        self.terminal = sys.stdout
        self.log = open(save_dir + "logfile.log", "a")

    def write(self, message):
        self.terminal.write(message)
        self.log.write(message)

    def flush(self):
        self.terminal.flush()
        self.log.flush()


def _natkey(s):
    return [int(t) if t.isdigit() else t for t in re.split(r"(\d+)", s)]


def save_checkpoint(state, save_dir="models", filename="checkpoint.pth.tar", max_model_num=8):
    """keep at most 8 files, dropping the naturally-sorted first = lowest Dice (train.py:171-176)"""
    torch.save(state, save_dir + filename)
    model_lists = sorted(glob.glob(save_dir + "*"), key=_natkey)
    while len(model_lists) > max_model_num:
        os.remove(model_lists[0])
        model_lists = sorted(glob.glob(save_dir + "*"), key=_natkey)


def latest_checkpoint(model_dir):
    """the file the reference resumes from: the naturally-sorted LAST one = highest Dice (train.py:83)"""
    files = sorted(os.listdir(model_dir), key=_natkey)
    if not files:
        raise RuntimeError(f"--cont-training: no checkpoint in {model_dir}")
    return os.path.join(model_dir, files[-1])


def main(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument("--train-dir", default="/LPBA_path/Train/")
    ap.add_argument("--val-dir", default="/LPBA_path/Val/")
    ap.add_argument("--synthetic", type=int, default=0, help="use N seeded synthetic subjects instead of .pkl files")
    ap.add_argument("--img-size", default="160,192,160")
    ap.add_argument("--max-epoch", type=int, default=30)
    ap.add_argument("--max-iters", type=int, default=0, help="stop each epoch after this many iterations (0 = all)")
    ap.add_argument("--lr", type=float, default=0.0001)
    ap.add_argument("--out", default=".")
    ap.add_argument("--cont-training", action="store_true", help="resume from the best checkpoint (train.py:61,:80-85)")
    ap.add_argument("--epoch-start", type=int, default=0, help="first epoch of a resumed run (train.py:59)")
    ap.add_argument("--no-restore-optimizer", action="store_true",
                    help="resume exactly like the reference: weights only, Adam restarts from zero moments (train.py:84)")
    ap.add_argument("--no-graph", action="store_true", help="issue every kernel of a step from Python instead of replaying a hipGraph")
    ap.add_argument("--overlap-allreduce", action="store_true",
                    help="multi-GPU: all-reduce the gradients in three buckets beside the backward (three hipGraph segments)")
    ap.add_argument("--host-loader", action="store_true",
                    help="read the .pkl pair from the host every iteration instead of caching all subjects in HBM")
    args = ap.parse_args(argv)
    same_seeds(24)
    rank, local, world = init_from_env()
    torch.cuda.set_device(local)

    weights = [1, 1]
    head_dim, num_heads = 6, [8, 4, 2, 1, 1]
    img_size = tuple(int(s) for s in args.img_size.split(","))
    save_dir = "modet-heads({}{}{}{}{})-rpe_headim_{}_ncc_{}_reg_{}_lr_{}_54r/".format(*num_heads, head_dim, weights[0],
                                                                                       weights[1], args.lr)
    exp_dir, log_dir = os.path.join(args.out, "experiments/" + save_dir), os.path.join(args.out, "logs/" + save_dir)
    f = None
    if rank == 0:
        os.makedirs(exp_dir, exist_ok=True)
        os.makedirs(log_dir, exist_ok=True)
        sys.stdout = Logger(log_dir)
        f = open(os.path.join(log_dir, "losses and dice.txt"), "a")

    model = ModeT(img_size, head_dim=head_dim, num_heads=num_heads, scale=1).cuda()
    best_dsc = 0
    resume = None
    if args.cont_training:
        ck = latest_checkpoint(exp_dir)
        # a checkpoint the reference wrote holds numpy scalars (param_groups[0]['lr'] = round(INIT_LR * np.power(..), 8),
        # train.py:166-168), which torch >= 2.6's default weights_only=True unpickler rejects: these are the user's own files
        resume = torch.load(ck, map_location="cpu", weights_only=False)
        model.load_state_dict(resume["state_dict"])
        if rank == 0:
            print(ck)
    trainer = Trainer(model, lr=args.lr, max_epoch=args.max_epoch, weights=weights,   # Adam(amsgrad) + NCC + Grad3d('l2')
                      overlap_allreduce=args.overlap_allreduce)
    if resume is not None and not args.no_restore_optimizer and isinstance(resume.get("optimizer"), dict) \
            and "state" in resume["optimizer"]:
        # the reference saves optimizer.state_dict() but never loads it back (train.py:80-85): a resumed run restarts
        # Adam's moments from zero.  We restore them, so save -> resume -> next step is identical to never stopping.
        trainer.load_state_dict(resume["optimizer"])
    if resume is not None:
        best_dsc = float(resume.get("best_dsc", 0))         # with or without the optimizer state

    if args.synthetic:
        train_set = data.SyntheticPairs(img_size, args.synthetic, 24)
        val_set = data.SyntheticPairs(img_size, max(2, args.synthetic // 2), 124, with_labels=True)
    else:
        train_set = data.LPBABrainDatasetS2S(glob.glob(args.train_dir + "*.pkl"))
        val_set = data.LPBABrainInferDatasetS2S(glob.glob(args.val_dir + "*.pkl"))
    # every subject resident in HBM, pairs indexed there; rank 0 alone validates
    train_cache = None if args.host_loader else data.DeviceVolumeCache(train_set)
    val_cache = None if (args.host_loader or rank != 0) else data.DeviceVolumeCache(val_set, with_labels=True)

    def train_pair(i):
        if train_cache is not None:
            return train_cache.pair(i)
        x, y = train_set[i][:2]
        return x[None].pin_memory().cuda(non_blocking=True), y[None].pin_memory().cuda(non_blocking=True)

    def val_pairs():
        for i in range(len(val_set)):
            if val_cache is not None:
                yield val_cache.pair(i)
            else:
                yield tuple(t[None].cuda() for t in val_set[i])

    graph_failed = False
    for epoch in range(args.epoch_start, args.max_epoch):
        if rank == 0:
            print("Training Starts")
        loss_all = utils.AverageMeter()
        order = np.random.RandomState(24 + epoch).permutation(len(train_set))      # same shuffle on every rank
        # identical step count on every rank (wrap-around padding): each step holds a blocking all-reduce
        mine = [int(order[i]) for i in lockstep_pairs_for_rank(len(order), rank, world)]
        n_iter = len(mine) if not args.max_iters else min(len(mine), args.max_iters)
        for idx in range(1, n_iter + 1):
            x, y = train_pair(mine[idx - 1])
            if not args.no_graph and trainer._graph is None and not graph_failed:
                try:
                    trainer.capture(x, y)           # forward+backward as one hipGraph from here on (engine.Trainer.capture)
                except RuntimeError as e:           # keep training eagerly (every rank takes the same branch below)
                    trainer.release_graph()
                    graph_failed = True
                    print(f"rank {rank}: hipGraph capture failed, continuing eagerly: {e}", file=sys.stderr)
                if world > 1:                       # all ranks replay or none does: the eager all-reduce pairs up either way,
                    ok = torch.tensor([0 if graph_failed else 1], device="cuda")          # but keep the modes aligned
                    torch.distributed.all_reduce(ok, op=torch.distributed.ReduceOp.MIN)
                    if int(ok.item()) == 0 and not graph_failed:
                        trainer.release_graph()
                        graph_failed = True
            loss, sim, reg = trainer.train_step(x, y, epoch=epoch)      # lr = poly_lr(epoch) inside (train.py:117)
            if rank == 0:
                lv = loss.item()                                        # one host sync per iteration, as train.py:130
                loss_all.update(lv, y.numel())
                print("Iter {} of {} loss {:.4f}, Img Sim: {:.6f}, Reg: {:.6f}".format(idx, n_iter, lv, sim.item(), reg.item()))
        if rank == 0:
            print("{} Epoch {} loss {:.4f}".format(save_dir, epoch, loss_all.avg))
            print("Epoch {} loss {:.4f}".format(epoch, loss_all.avg), file=f, end=" ")
            eval_dsc = utils.AverageMeter()
            with torch.no_grad():
                model.eval()
                for x, y, x_seg, y_seg in val_pairs():
                    _, flow = model(x, y)
                    _, dsc = utils.warp_labels_and_dice(x_seg, flow, y_seg)     # fused GPU eval tail (train.py:152-153)
                    eval_dsc.update(dsc, x.size(0))
                    print(epoch, ":", eval_dsc.avg)
            best_dsc = max(eval_dsc.avg, best_dsc)
            print(eval_dsc.avg, file=f)
            f.flush()
            save_checkpoint({"epoch": epoch + 1, "state_dict": model.state_dict(), "best_dsc": best_dsc,
                             "optimizer": trainer.state_dict()},
                            save_dir=exp_dir, filename="dsc{:.3f}.pth.tar".format(eval_dsc.avg))
        if world > 1:
            torch.distributed.barrier()             # nobody starts the next epoch's all-reduces while rank 0 validates
    return best_dsc


if __name__ == "__main__":
    main()
