"""Paper-faithful loss variants: the losses as the reference's Caffe training graph defines them
(``experiments/depth_odometry_feature/train.prototxt``), which differ from ``pytorch_version/loss_functions.py`` in
three ways:

  * ``AbsLoss`` (caffe/src/caffe/layers/abs_loss_layer.cu:10-34) is a PER-SAMPLE SUM (sum |a - b| / batch), has no
    exact-zero validity mask, and differentiates |d| as (d > 0) - (d <= 0);
  * the smoothness term is FIRST-order and EDGE-AWARE on the inverse depth: central differences weighted by
    exp(-0.33 * sum_c |dI_c|) (train.prototxt:4452-4661, fillers caffe/include/caffe/filler.hpp:267-316), weight 10;
  * the feature extractor is frozen (train.prototxt:4869-: ``lr_mult: 0``), so the feature term sends no gradient into
    the feature maps.

These functions have no counterpart in the reference's Python files and Caffe cannot run in the build image: parity
is UNPINNED -- the kernels are tested against ``oracle/``'s restatement of the layer sources and the oracle against
finite differences.  The geometry is the Caffe chain (se(3) exponential map, pixel-coordinate warp) of
``unsupervise_dvo.py``."""
import torch

from dvf import lib as _L
from dvf.ops import EdgeSmoothLossFn, PhotoLossFn


def abs_warp_loss(tgt, srcs, depth, poses_se3, intrinsics, intrinsics_inv, img_scale=1.0):
    """sum over the views of AbsLoss(warp(src_v), tgt): ``warp_error_LR`` + ``warp_error_R12``
    (train.prototxt:4428-4446) when called with (left image, previous right image) and (T_R2L, T_2to1), and
    ``warp_feat_error`` (:5571-5588) on feature maps.  poses_se3: list of [B,6] se(3) vectors (w, u)."""
    pose = torch.stack(list(poses_se3), dim=0)
    flags = _L.POSE_SE3 | _L.PIXEL_COORDS | _L.CAFFE_ABSLOSS
    d = depth[:, 0] if depth.dim() == 4 else depth
    return PhotoLossFn.apply(tgt, d, pose, intrinsics, intrinsics_inv, None, (flags, float(img_scale)), *srcs)


def edge_aware_smooth_loss(inv_depth, img, img_scale=1.0, edge_k=0.33, weight=1.0):
    """``smoothness1`` + ``smoothness2`` of train.prototxt:4452-4661 (the graph applies loss_weight 10 to each)."""
    return EdgeSmoothLossFn.apply(inv_depth, img, float(img_scale), float(edge_k), float(weight))
