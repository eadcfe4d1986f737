"""Drop-in for the reference's ``pytorch_version/inverse_warp.py`` on MI355X.

Same public names, argument order, defaults and assertion messages; the whole
pixel2cam -> pose_vec2mat -> K@[R|t] -> cam2pixel -> grid_sample chain runs as ONE hand-written HIP
kernel (forward) and one (backward) from ``libdvf_hip.so``.  ``align_corners`` is an extension: the
default ``False`` is what the reference executes under current torch (SURVEY.md preamble #5).
"""
import torch

from dvf import lib as _L
from dvf.ops import InverseWarpFn


def check_sizes(input, input_name, expected):
    """Same contract as the reference's check_sizes (inverse_warp.py:18-23): digits are exact sizes,
    letters are free dimensions."""
    ok = input.ndimension() == len(expected) and all(
        (not s.isdigit()) or input.size(i) == int(s) for i, s in enumerate(expected))
    assert ok, "wrong size for {}, expected {}, got  {}".format(input_name, 'x'.join(expected), list(input.size()))


def inverse_warp(img, depth, pose, intrinsics, intrinsics_inv, rotation_mode='euler', padding_mode='zeros',
                 align_corners=False):
    """Inverse warp a source image to the target image plane.

    img [B,C,H,W] (any C: the reference's loss_functions.py copy drops the B3HW check so that
    32-channel feature maps can be warped), depth [B,H,W], pose [B,6] = (tx,ty,tz,rx,ry,rz) target->source,
    intrinsics / intrinsics_inv [B,3,3].  Returns the warped source, [B,C,H,W]."""
    check_sizes(img, 'img', 'BCHW')
    check_sizes(depth, 'depth', 'BHW')
    check_sizes(pose, 'pose', 'B6')
    check_sizes(intrinsics, 'intrinsics', 'B33')
    check_sizes(intrinsics_inv, 'intrinsics', 'B33')
    assert(intrinsics_inv.size() == intrinsics.size())
    flags = _L.geom_flags(rotation_mode, padding_mode, align_corners)
    return InverseWarpFn.apply(img, depth, pose, intrinsics, intrinsics_inv, flags)
