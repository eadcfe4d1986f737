"""Drop-in for the reference's ``pytorch_version/loss_functions_sfm.py`` (multi-scale API of ``train.py``)."""
import torch

from dvf import lib as _L
from dvf.conv import area_downsample
from dvf.ops import ExplainabilityLossFn, PhotoLossFn, SmoothLossFn


def photometric_reconstruction_loss(tgt_img, ref_imgs, intrinsics, intrinsics_inv, depth, explainability_mask, pose,
                                    rotation_mode='euler', padding_mode='zeros', align_corners=False,
                                    check_nan=False):
    """Sum over scales and reference views of the masked, explainability-weighted L1 between the target and
    the inverse-warped references (reference loss_functions_sfm.py:9-46).  One fused kernel per scale handles
    all views.  The reference's per-view NaN assert (:34) forces a device sync per view per scale; here it is
    opt-in (``check_nan=True``) and done once on the summed loss."""
    if type(explainability_mask) not in [tuple, list]:
        explainability_mask = [explainability_mask]
    if type(depth) not in [list, tuple]:
        depth = [depth]
    assert(pose.size(1) == len(ref_imgs))
    flags = _L.geom_flags(rotation_mode, padding_mode, align_corners)
    pose_vb6 = pose.transpose(0, 1).contiguous()                        # [V, B, 6]
    loss = 0
    for d, mask in zip(depth, explainability_mask):
        assert(mask is None or d.size()[2:] == mask.size()[2:])
        b, _, h, w = d.size()
        downscale = tgt_img.size(2) / h
        tgt_s = area_downsample(tgt_img, (h, w))                         # :18
        refs_s = [area_downsample(r, (h, w)) for r in ref_imgs]          # :19
        k_s = torch.cat((intrinsics[:, 0:2] / downscale, intrinsics[:, 2:]), dim=1)               # :20
        kinv_s = torch.cat((intrinsics_inv[:, :, 0:2] * downscale, intrinsics_inv[:, :, 2:]), dim=2)   # :21
        loss = loss + PhotoLossFn.apply(tgt_s, d[:, 0], pose_vb6, k_s, kinv_s, mask, flags, *refs_s)
    if check_nan:
        assert((loss == loss).item() == 1)
    return loss


def explainability_loss(mask):
    """sum_s BCE(mask_s, 1)  (reference loss_functions_sfm.py:49-56)."""
    if type(mask) not in [tuple, list]:
        mask = [mask]
    return ExplainabilityLossFn.apply(*mask)


def smooth_loss(pred_map, scale_factor):
    """Second-order smoothness over a list of maps, weight divided by ``scale_factor`` per scale (:59-77)."""
    if type(pred_map) not in [tuple, list]:
        pred_map = [pred_map]
    return SmoothLossFn.apply(float(scale_factor), *pred_map)


@torch.no_grad()
def compute_errors(gt, pred, crop=True):
    """Evaluation-only depth metrics, same contract as the reference helper (loss_functions_sfm.py:80-116):
    per sample, valid = 0 < gt < 80 (inside the Garg/Eigen crop when ``crop``), prediction clamped to
    [1e-3, 80] and median-scaled to the ground truth; returns batch means of
    [abs_diff, abs_rel, sq_rel, a1, a2, a3].  Plain tensor code, not on the training hot path."""
    n, h, w = gt.shape
    region = torch.ones(h, w, dtype=torch.bool, device=gt.device)
    if crop:
        region[:] = False
        region[int(0.40810811 * h):int(0.99189189 * h), int(0.03594771 * w):int(0.96405229 * w)] = True
    totals = torch.zeros(6, dtype=torch.float64)
    for g_i, p_i in zip(gt, pred):
        keep = (g_i > 0) & (g_i < 80) & region
        g, p = g_i[keep], p_i[keep].clamp(1e-3, 80)
        p = p * (g.median() / p.median())
        ratio = torch.maximum(g / p, p / g)
        err = (g - p).abs()
        row = [err.mean(), (err / g).mean(), (err * err / g).mean()] + [(ratio < 1.25 ** e).float().mean() for e in (1, 2, 3)]
        totals += torch.stack([r.double().cpu() for r in row])
    return (totals / n).tolist()
