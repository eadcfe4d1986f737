#!/usr/bin/env python3
"""MI355X drop-in for the reference's ``pytorch_version/unsupervise.py``: depth (DispNetS) + odometry (PoseExpNet)
[+ FeatExtractor] trained with image reconstruction + 0.1 * feature reconstruction + 10 * smoothness
(unsupervise.py:101-111), Adam(lr 1e-3, weight_decay 1e-8) (:241).

Same flags and defaults as the reference (unsupervise.py:35-57).  The reference's odometry net is the 8-bit
fixed-point ``FixOdometryNet`` (out of scope); as BASELINE.json's north_star specifies, PoseExpNet
takes its place.  Data is the seeded synthetic stream; ``--features`` adds the FeatExtractor term (cfg 3).
"""
import argparse
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))

import torch  # noqa: E402

from DispNetS import DispNetS  # noqa: E402
from PoseExpNet import PoseExpNet  # noqa: E402
from dvf import cli  # noqa: E402
from dvf.steps import unsupervise_losses  # noqa: E402

parser = argparse.ArgumentParser(formatter_class=argparse.ArgumentDefaultsHelpFormatter)
parser.add_argument("--odometry", default=None, type=str, help="checkpoint of the odometry (pose) net")
parser.add_argument("--depth", default=None, type=str, help="checkpoint of the depth net")
parser.add_argument("--epochs", type=int, default=50, metavar="N")
parser.add_argument("--lr", type=float, default=1e-3, metavar="LR")
parser.add_argument("--momentum", type=float, default=0.9, metavar="M")
parser.add_argument("--weight-decay", type=float, default=1e-8, metavar="WD")
parser.add_argument("--seed", type=int, default=2019, metavar="S")
parser.add_argument("-b", "--batch-size", default=64, type=int, help="mini-batch size PER GPU")
parser.add_argument("-g", "--gpu-id", type=int, metavar="N", default=-1)
parser.add_argument("--dataset-dir", default="/home/share/kitti_odometry/dataset/", type=str)
parser.add_argument("--train-sequences", default=["01", "02", "03", "04", "05", "06", "07", "08", "09", "10"], type=str, nargs="*")
parser.add_argument("--test-sequences", default=["00"], type=str, nargs="*")
parser.add_argument("-j", "--workers", default=4, type=int, metavar="N")
parser.add_argument("--log-interval", type=int, default=10, metavar="N")
parser.add_argument("--output-dir", type=str, default="./checkpoints")
parser.add_argument("--features", action="store_true", help="add the 0.1 * feature-reconstruction term (FeatExtractor)")
cli.add_common_flags(parser)


def main():
    args = parser.parse_args()
    args._rank, args._world, args._device = cli.init_distributed()
    torch.manual_seed(args.seed)
    depth_net, odometry_net = DispNetS(), PoseExpNet(output_exp=True)
    if args.odometry:
        cli.load_pretrained(odometry_net, args.odometry)
    else:
        odometry_net.init_weights()
    if args.depth:
        cli.load_pretrained(depth_net, args.depth)
    else:
        depth_net.init_weights()
    nets = [odometry_net, depth_net]                                   # optimizer group order of unsupervise.py:234-238
    ckpts = [("best_vo_checkpoint.pth.tar", odometry_net), ("best_depth_checkpoint.pth.tar", depth_net)]
    feat = None
    if args.features:
        from feat_extractor import FeatExtractor
        feat = FeatExtractor()
        feat.init_weights()
        nets.append(feat)
        ckpts.append(("best_feat_checkpoint.pth.tar", feat))
    for n in nets:
        n.to(args._device).train()

    def loss_fn(batch):
        return unsupervise_losses(depth_net, odometry_net, batch, feat_extractor=feat)

    terms = ["total", "img", "smooth"] + (["feat"] if feat is not None else [])
    cli.run_training(args, nets, loss_fn, args.lr, (0.9, 0.999), args.weight_decay, terms, ckpts)


if __name__ == "__main__":
    main()
