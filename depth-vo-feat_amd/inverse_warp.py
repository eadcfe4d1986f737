"""Drop-in for the reference's ``pytorch_version/inverse_warp.py`` on MI355X.

Same public names, argument order, defaults and assertion messages; the whole
pixel2cam -> pose_vec2mat -> K@[R|t] -> cam2pixel -> grid_sample chain runs as ONE hand-written HIP
kernel (forward) and one (backward) from ``libdvf_hip.so``.  ``align_corners`` is an extension: the
default ``False`` is what the reference executes under current torch (SURVEY.md preamble #5).
"""
import torch

from dvf import lib as _L
from dvf.ops import Cam2PixelFn, InverseWarpFn, Pixel2CamFn, PoseVec2MatFn

pixel_coords = None      # module-level cache like the reference's (inverse_warp.py:5); only set_id_grid() fills it


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


def set_id_grid(depth):
    """Cache the [1,3,H,W] grid of (u, v, 1) pixel coordinates in ``pixel_coords`` (reference :8-15).  The fused
    kernels never read it (they generate u, v from the thread index); it exists for callers that use the global."""
    global pixel_coords
    b, h, w = depth.size()
    v = torch.arange(0, h, device=depth.device).view(1, h, 1).expand(1, h, w).type_as(depth)
    u = torch.arange(0, w, device=depth.device).view(1, 1, w).expand(1, h, w).type_as(depth)
    pixel_coords = torch.stack((u, v, torch.ones(1, h, w, device=depth.device).type_as(depth)), dim=1)


def pixel2cam(depth, intrinsics_inv):
    """depth [B,H,W], intrinsics_inv [B,3,3] -> camera-frame points [B,3,H,W] (reference :26-40)."""
    check_sizes(depth, 'depth', 'BHW')
    return Pixel2CamFn.apply(depth, intrinsics_inv)


def cam2pixel(cam_coords, proj_c2p_rot, proj_c2p_tr, padding_mode):
    """Camera-frame points [B,3,H,W] -> normalised sampling grid [B,H,W,2] in [-1,1] (reference :43-74);
    with padding_mode='zeros' coordinates outside the image are overwritten with 2."""
    return Cam2PixelFn.apply(cam_coords, proj_c2p_rot, proj_c2p_tr, _L.geom_flags('euler', padding_mode))


def pose_vec2mat(vec, rotation_mode='euler'):
    """6-DoF vector (tx,ty,tz,rx,ry,rz) [B,6] -> [R|t] [B,3,4] (reference :141-157)."""
    return PoseVec2MatFn.apply(vec, _L.geom_flags(rotation_mode))


def euler2mat(angle):
    """Euler angles [B,3] -> R = Rx @ Ry @ Rz [B,3,3] (reference :77-114)."""
    vec = torch.cat((torch.zeros_like(angle), angle), dim=1)
    return PoseVec2MatFn.apply(vec, _L.geom_flags('euler'))[:, :, :3]


def quat2mat(quat):
    """Last three coefficients of a (1, q) quaternion [B,3] -> rotation matrix [B,3,3] (reference :117-138)."""
    vec = torch.cat((torch.zeros_like(quat), quat), dim=1)
    return PoseVec2MatFn.apply(vec, _L.geom_flags('quat'))[:, :, :3]
