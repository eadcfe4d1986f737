#!/usr/bin/env python3
"""MI355X drop-in for the reference's ``pytorch_version/unsupervise_dvo.py``: depth (DispNetS) + odometry trained
through the Caffe-style geometry chain -- se(3) exponential map, GeoTransform, PinHole projection, inverse warping
in pixel coordinates -- with stereo + temporal masked L1 and 10 * smoothness (unsupervise_dvo.py:83-122).

The reference script cannot run (its geo_transform() hits exit(0), geo_transform.py:31); its intended dataflow is
defined by the Caffe layers, which is what the fused kernel's DVF_POSE_SE3 | DVF_PIXEL_COORDS front
end implements.  Same flags as unsupervise.py; PoseExpNet replaces the fixed-point FixOdometryNet."""
import argparse
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))

import torch  # noqa: E402

from DispNetS import DispNetS  # noqa: E402
from PoseExpNet import PoseExpNet  # noqa: E402
from dvf import cli  # noqa: E402
from dvf.steps import unsupervise_dvo_losses  # noqa: E402

parser = argparse.ArgumentParser(formatter_class=argparse.ArgumentDefaultsHelpFormatter)
parser.add_argument("--odometry", default=None, type=str)
parser.add_argument("--depth", default=None, type=str)
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
    for n in (odometry_net, depth_net):
        n.to(args._device).train()

    def loss_fn(batch):
        # this script's chain consumes the stereo pose in se(3) order (w, u), exactly as the dataset files hold it
        # (data/dataset_builder.py:155); both the synthetic stream and un_dataset.to_batch provide it under this key
        b = dict(batch)
        b["T_R2L"] = batch["T_R2L_se3"]
        return unsupervise_dvo_losses(depth_net, odometry_net, b)

    cli.run_training(args, [odometry_net, depth_net], loss_fn, args.lr, (0.9, 0.999), args.weight_decay,
                     ["total", "photo", "smooth"],
                     [("best_vo_checkpoint.pth.tar", odometry_net), ("best_depth_checkpoint.pth.tar", depth_net)])


if __name__ == "__main__":
    main()
