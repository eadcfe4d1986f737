"""Drop-in for the reference's ``pytorch_version/loss_functions.py`` (single-scale API of
``unsupervise.py``) on MI355X: each loss is a fused HIP kernel pair (forward / backward)."""
import torch

from dvf import lib as _L
from dvf.ops import PhotoLossFn, SmoothLossFn
from inverse_warp import inverse_warp, check_sizes  # noqa: F401  (re-exported like the reference's copy)


def photometric_reconstruction_loss(img_R2, img_R1, img_L2, depth, T_2to1, T_R2L, intrinsics, intrinsics_inv,
                                    rotation_mode='euler', padding_mode='zeros', align_corners=False, img_scale=1.0):
    """Temporal (img_R1 warped with T_2to1) + stereo (img_L2 warped with T_R2L) masked L1 against img_R2
    (reference loss_functions.py:7-20).  Works for images (C=3) and 32-channel feature maps alike; grads
    flow to depth, both poses and any image/feature input that requires them.
    ``img_scale`` (extension, default 1): the three images are used as img_scale * img inside the kernel -- the
    reference's call site passes 0.004 * img (unsupervise.py:101); this gives the same bits without the three
    scaled copies."""
    assert(intrinsics_inv.size() == intrinsics.size())
    flags = _L.geom_flags(rotation_mode, padding_mode, align_corners)
    pose = torch.stack((T_2to1, T_R2L), dim=0)                 # [V=2, B, 6]
    if img_scale != 1.0:
        flags = (flags, float(img_scale))
    return PhotoLossFn.apply(img_R2, depth, pose, intrinsics, intrinsics_inv, None, flags, img_R1, img_L2)


def smooth_loss(pred_map, scale_factor=1):
    """Second-order smoothness of one map or a list of maps (reference loss_functions.py:23-41)."""
    if type(pred_map) not in [tuple, list]:
        pred_map = [pred_map]
    return SmoothLossFn.apply(float(scale_factor), *pred_map)
