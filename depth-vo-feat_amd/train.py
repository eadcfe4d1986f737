#!/usr/bin/env python3
"""MI355X drop-in for the reference's ``pytorch_version/train.py``: DispNetS + PoseExpNet(sfm) trained with the
4-scale photometric loss (explainability masks), explainability regulariser, smoothness and the stereo-pose MSE.

Same flags and defaults as the reference (train.py:31-67).  Differences, all forced by scope:
data is the seeded synthetic stream unless --data-root is given, multi-GPU is one process per GPU with RCCL all-reduce
instead of nn.DataParallel.  ``validate()`` (odometry L1 against the ground-truth relative poses of the test sequences,
train.py:220-247) runs after every epoch when --val-root points at a KITTI-odometry tree (sequences/<seq>/image_2,
poses/<seq>.txt) and then decides the best checkpoint like the reference; otherwise the best training loss does.

    python train.py -b 4 --epochs 2                       # 1 GPU
    python -m torch.distributed.run --nproc-per-node 8 --master-addr 127.0.0.1 train.py -b 4
"""
import argparse
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))

import torch  # noqa: E402

from DispNetS import DispNetS  # noqa: E402
from PoseExpNet_sfm import PoseExpNet  # noqa: E402
from dvf import cli  # noqa: E402
from dvf.steps import train_sfm_losses  # noqa: E402

parser = argparse.ArgumentParser(description="Unsupervised depth + pose training (sfm loss form) on MI355X",
                                 formatter_class=argparse.ArgumentDefaultsHelpFormatter)
parser.add_argument("--dataset-dir", default="/home/share/kitti_odometry/dataset/", type=str, help="accepted for compatibility; unused (synthetic data)")
parser.add_argument("--train-sequences", default=["01", "02", "03", "04", "05", "06", "07", "08", "09", "10"], type=str, nargs="*")
parser.add_argument("--test-sequences", default=["00"], type=str, nargs="*")
parser.add_argument("--rotation-mode", type=str, choices=["euler", "quat"], default="euler")
parser.add_argument("--padding-mode", type=str, choices=["zeros", "border"], default="zeros")
parser.add_argument("-j", "--workers", default=4, type=int, metavar="N")
parser.add_argument("--epochs", default=200, type=int, metavar="N")
parser.add_argument("--epoch-size", default=0, type=int, metavar="N", help="iterations per epoch (0: --steps-per-epoch)")
parser.add_argument("-b", "--batch-size", default=4, type=int, metavar="N", help="mini-batch size PER GPU")
parser.add_argument("--lr", "--learning-rate", default=2e-4, type=float, metavar="LR")
parser.add_argument("--momentum", default=0.9, type=float, metavar="M", help="alpha parameter for adam")
parser.add_argument("--beta", default=0.999, type=float, metavar="M", help="beta parameter for adam")
parser.add_argument("--weight-decay", "--wd", default=0, type=float, metavar="W")
parser.add_argument("--pretrained-disp", dest="pretrained_disp", default=None, metavar="PATH")
parser.add_argument("--pretrained-exppose", dest="pretrained_exp_pose", default=None, metavar="PATH")
parser.add_argument("--seed", default=0, type=int)
parser.add_argument("-p", "--photo-loss-weight", type=float, metavar="W", default=1)
parser.add_argument("-m", "--mask-loss-weight", type=float, metavar="W", default=0)
parser.add_argument("-s", "--smooth-loss-weight", type=float, metavar="W", default=0.1)
parser.add_argument("--smooth-loss-factor", type=float, metavar="W", default=2)
parser.add_argument("-g", "--gpu-id", type=int, metavar="N", default=-1, help="accepted for compatibility; ranks pick LOCAL_RANK")
parser.add_argument("--output-dir", type=str, default="./checkpoints")
parser.add_argument("--nb-ref-imgs", type=int, default=2, help="reference views (2 = temporal + stereo; 4 = 5-frame window)")
parser.add_argument("--val-root", default=None, help="KITTI odometry tree for validate() on --test-sequences (train.py:220-247)")
cli.add_common_flags(parser)


def main():
    args = parser.parse_args()
    args._rank, args._world, args._device = cli.init_distributed()
    if args.epoch_size:
        args.steps_per_epoch = args.epoch_size
    torch.manual_seed(args.seed)
    disp_net = DispNetS()
    pose_exp_net = PoseExpNet(nb_ref_imgs=args.nb_ref_imgs, output_exp=True)      # train.py:126-127
    if args.pretrained_exp_pose:
        cli.load_pretrained(pose_exp_net, args.pretrained_exp_pose, strict=False)
    else:
        pose_exp_net.init_weights()
    if args.pretrained_disp:
        cli.load_pretrained(disp_net, args.pretrained_disp)
    else:
        disp_net.init_weights()
    disp_net.to(args._device).train()
    pose_exp_net.to(args._device).train()

    def loss_fn(batch):
        return train_sfm_losses(disp_net, pose_exp_net, batch, args.photo_loss_weight, args.mask_loss_weight,
                                args.smooth_loss_weight, args.smooth_loss_factor, args.rotation_mode, args.padding_mode)

    terms = ["total", "photo", "smooth", "lr"] + (["exp"] if args.mask_loss_weight > 0 else [])
    val_fn = None
    if args.val_root and args._rank == 0:
        from dataset import pose_framework_KITTI
        val_set = pose_framework_KITTI(args.val_root, args.test_sequences, img_height=args.height, img_width=args.width,
                                       shuffle=False)
        val_loader = torch.utils.data.DataLoader(val_set, batch_size=args.batch_size, shuffle=False, num_workers=0)
        val_fn = lambda: cli.validate(pose_exp_net, val_loader, args._device)      # noqa: E731
    cli.run_training(args, [disp_net, pose_exp_net], loss_fn, args.lr, (args.momentum, args.beta), args.weight_decay,
                     terms, [("best_vo_checkpoint.pth.tar", pose_exp_net), ("best_depth_checkpoint.pth.tar", disp_net)],
                     n_views=max(2, args.nb_ref_imgs), val_fn=val_fn)


if __name__ == "__main__":
    main()
