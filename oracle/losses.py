"""Oracle (test infrastructure): photometric / feature-reconstruction, smoothness and
explainability losses.

Restates reference ``pytorch_version/loss_functions.py:7-41`` (single-scale API used by
``unsupervise.py``) and ``pytorch_version/loss_functions_sfm.py:9-77`` (4-scale API used by
``train.py``).  There is no SSIM and no separate feature-reconstruction function in the
reference: the feature term is the same photometric function on 32-channel maps
(``unsupervise.py:104-109``).
"""
import torch
import torch.nn.functional as F

from .geometry import inverse_warp


def _valid_mask(warped):
    """1 - prod_c [warped_c == 0]: a pixel is masked only when ALL channels are exactly 0.
    loss_functions.py:11,16 ; loss_functions_sfm.py:27."""
    return 1 - (warped == 0).prod(1, keepdim=True).type_as(warped)


def photometric_reconstruction_loss(img_R2, img_R1, img_L2, depth, T_2to1, T_R2L, intrinsics,
                                    intrinsics_inv, rotation_mode="euler", padding_mode="zeros",
                                    align_corners=False):
    """Temporal (R1 with T_2to1) + stereo (L2 with T_R2L) masked L1, each a mean over B*C*H*W
    (masked pixels stay in the denominator).  loss_functions.py:7-20."""
    loss = 0
    for src, pose in ((img_R1, T_2to1), (img_L2, T_R2L)):
        warped = inverse_warp(src, depth, pose, intrinsics, intrinsics_inv, rotation_mode,
                              padding_mode, align_corners)
        loss = loss + ((img_R2 - warped) * _valid_mask(warped)).abs().mean()
    return loss


def _second_order_terms(m):
    """dx2, dxdy, dydx, dy2 of one [B,C,H,W] map.  loss_functions.py:24-38."""
    dy = m[:, :, 1:] - m[:, :, :-1]
    dx = m[:, :, :, 1:] - m[:, :, :, :-1]
    dx2 = dx[:, :, :, 1:] - dx[:, :, :, :-1]
    dxdy = dx[:, :, 1:] - dx[:, :, :-1]
    dydx = dy[:, :, :, 1:] - dy[:, :, :, :-1]
    dy2 = dy[:, :, 1:] - dy[:, :, :-1]
    return dx2, dxdy, dydx, dy2


def smooth_loss(pred_map, scale_factor=1):
    """sum_s w_s * (mean|dx2| + mean|dxdy| + mean|dydx| + mean|dy2|), w_0 = 1, w_{s+1} = w_s / f.
    loss_functions.py:23-41 (default f=1) ; loss_functions_sfm.py:59-77 (f passed, train.py uses 2)."""
    if type(pred_map) not in (tuple, list):
        pred_map = [pred_map]
    loss, weight = 0, 1.0
    for m in pred_map:
        loss = loss + sum(t.abs().mean() for t in _second_order_terms(m)) * weight
        weight /= scale_factor
    return loss


def photometric_reconstruction_loss_sfm(tgt_img, ref_imgs, intrinsics, intrinsics_inv, depth,
                                        explainability_mask, pose, rotation_mode="euler",
                                        padding_mode="zeros", align_corners=False):
    """Multi-scale form.  loss_functions_sfm.py:9-46.  Per scale: area-downsample target and
    references to the depth map's size (:18-19), K rows 0-1 divided and Kinv columns 0-1
    multiplied by the downscale (:20-21), per reference view: warp with pose[:, i], exact-zero
    mask, optional explainability mask multiply (:30-31), abs().mean(), NaN assert (:34)."""
    if type(explainability_mask) not in (tuple, list):
        explainability_mask = [explainability_mask]
    if type(depth) not in (tuple, list):
        depth = [depth]
    assert pose.shape[1] == len(ref_imgs)
    total = 0
    for d, mask in zip(depth, explainability_mask):
        b, _, h, w = d.shape
        down = tgt_img.shape[2] / h
        tgt_s = F.interpolate(tgt_img, (h, w), mode="area")
        refs_s = [F.interpolate(r, (h, w), mode="area") for r in ref_imgs]
        k_s = torch.cat((intrinsics[:, 0:2] / down, intrinsics[:, 2:]), dim=1)
        kinv_s = torch.cat((intrinsics_inv[:, :, 0:2] * down, intrinsics_inv[:, :, 2:]), dim=2)
        for i, ref in enumerate(refs_s):
            warped = inverse_warp(ref, d[:, 0], pose[:, i], k_s, kinv_s, rotation_mode,
                                  padding_mode, align_corners)
            diff = (tgt_s - warped) * _valid_mask(warped)
            if mask is not None:
                diff = diff * mask[:, i:i + 1].expand_as(diff)
            total = total + diff.abs().mean()
            assert bool(total == total), "NaN in photometric loss"
    return total


def explainability_loss(mask):
    """sum_s BCE(mask_s, 1) = sum_s -mean(log mask_s) (torch clamps log at -100).
    loss_functions_sfm.py:49-56."""
    if type(mask) not in (tuple, list):
        mask = [mask]
    loss = 0
    for m in mask:
        loss = loss + (-torch.clamp(torch.log(m), min=-100.0)).mean()
    return loss


# ----------------------------------------------------------------------------- paper-faithful variants (Caffe graph)
class _CaffeAbsSum(torch.autograd.Function):
    """sum |v| with AbsLoss's backward convention: d|v|/dv = (v > 0) - (v <= 0) applied to diff = bottom0 - bottom1
    (caffe/src/caffe/layers/abs_loss_layer.cu:28-34) -- an exact zero gets a gradient of -1 w.r.t. `diff`, not 0."""

    @staticmethod
    def forward(ctx, diff):
        ctx.save_for_backward(diff)
        return diff.abs().sum()

    @staticmethod
    def backward(ctx, g):
        (diff,) = ctx.saved_tensors
        return g * torch.where(diff > 0, torch.ones_like(diff), -torch.ones_like(diff))


def abs_loss_caffe(bottom0, bottom1):
    """Caffe AbsLoss (abs_loss_layer.cu:10-26): sum |bottom0 - bottom1| / N with N = bottom0.num() (the batch size) --
    a per-sample SUM, not a mean -- and the sign convention above.  No validity mask: the Caffe graph has none
    (experiments/depth_odometry_feature/train.prototxt:4428-4446)."""
    return _CaffeAbsSum.apply(bottom0 - bottom1) / bottom0.shape[0]


def edge_aware_smooth_caffe(inv_depth, img, edge_k=0.33):
    """Edge-aware first-order smoothness of the inverse depth, train.prototxt:4452-4661 (without its loss_weight 10):
    3x3 VALID cross-correlations with the EdgeX / EdgeY fillers (caffe/include/caffe/filler.hpp:267-316: EdgeX is the
    central difference along y, EdgeY along x, both with +-0.5), image edges |.| summed over channels with -0.33,
    exp(), product with the inverse-depth differences, AbsLoss against zeros (bottom0 = 0, bottom1 = dx)."""
    c = img.shape[1]
    ex = img.new_tensor([[0, -0.5, 0], [0, 0, 0], [0, 0.5, 0]]).view(1, 1, 3, 3)
    ey = img.new_tensor([[0, 0, 0], [-0.5, 0, 0.5], [0, 0, 0]]).view(1, 1, 3, 3)
    gx = torch.exp(-edge_k * F.conv2d(img, ex.expand(c, 1, 3, 3), groups=c).abs().sum(1, keepdim=True))
    gy = torch.exp(-edge_k * F.conv2d(img, ey.expand(c, 1, 3, 3), groups=c).abs().sum(1, keepdim=True))
    dx = gx * F.conv2d(inv_depth, ex)
    dy = gy * F.conv2d(inv_depth, ey)
    zeros = torch.zeros_like(dx)
    return abs_loss_caffe(zeros, dx) + abs_loss_caffe(zeros, dy)
