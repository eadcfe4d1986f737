"""The training-step bodies of the reference's entry scripts, on the HIP path.

  unsupervise_losses  unsupervise.py:83-111   img recon + 10*smooth (+ 0.1*feature recon), single scale
  train_sfm_losses    train.py:179-203        4-scale photometric with masks + w3*smooth + stereo-pose MSE
Each returns (total_loss, dict of detached per-term losses); the caller does zero_grad / backward / step.
No ``.item()`` here: the reference's 5 host syncs per step (train.py:205-209) are left to the logger."""

import torch

import loss_functions as LF
import loss_functions_sfm as LS
from dvf import lib as L
from dvf.conv import reciprocal


POSE_STREAM = True       # the pose network runs on the auxiliary stream beside the depth network


def _side_by_side(aux_fn, main_fn):
    """Run two independent sub-networks concurrently: ``aux_fn`` (the small pose network) on a second stream, ``main_fn``
    (the depth network) on the current one.  The deep layers of either network launch fewer blocks than the GPU has
    CUs, so the two fill each other's gaps; autograd runs each backward on its forward stream, so the backward passes
    overlap too.  FlatAdam.step() joins the auxiliary stream (dvf/lib.py AUX_STREAMS).  ``POSE_STREAM = False`` (module
    attribute; tools set it, nothing on the product path reads the environment) puts both networks on one stream."""
    if L.SERIALIZE or not POSE_STREAM:
        # same HOST order as below (depth network first): the order in which gradient buckets become ready -- and with it
        # the order of the all-reduce calls -- must not depend on this switch (bench.py times rank 0 serialised while the
        # other ranks run the overlapped schedule: mismatched collective orders would hang RCCL)
        m_out = main_fn()
        return aux_fn(), m_out
    cur = torch.cuda.current_stream()
    aux = L.aux_stream(cur.device)
    aux.wait_stream(cur)
    # Host order: the depth network first, the pose network LAST.  Autograd runs ready nodes latest-created first, so the
    # pose network's whole backward pass is enqueued (on its own stream) before the depth network's and overlaps it from
    # the start; the other order leaves it -- and its weight gradients, Adam and repack -- as a serial tail of the step.
    m_out = main_fn()
    with torch.cuda.stream(aux):
        a_out = aux_fn()
    cur.wait_stream(aux)

    def _mark(t):
        if torch.is_tensor(t):
            t.record_stream(cur)          # produced on aux, consumed by the losses on the current stream
        elif isinstance(t, (list, tuple)):
            for x in t:
                _mark(x)
    _mark(a_out)
    return a_out, m_out


def unsupervise_losses(depth_net, pose_net, batch, feat_extractor=None, img_scale=0.004, smooth_weight=10.0,
                       feat_weight=0.1, depth_eps=1e-4):
    R2, R1, L2 = batch["img_R2"], batch["img_R1"], batch["img_L2"]
    # unsupervise.py:92,95 (cat(R2, R1) done virtually) and :94 (intended: finest disparity)
    (_, T_2to1), disps = _side_by_side(lambda: pose_net((R2, R1)), lambda: depth_net(R2))
    inv_depth = disps[0]
    depth = reciprocal(inv_depth, depth_eps).squeeze(1)            # :99
    img_loss = LF.photometric_reconstruction_loss(R2, R1, L2, depth, T_2to1, batch["T_R2L"], batch["K"], batch["Kinv"],
                                                  img_scale=img_scale)   # :101 (0.004 * img applied inside the kernel)
    smooth = LF.smooth_loss(depth.unsqueeze(1))                    # :102
    terms = {"img": img_loss.detach(), "smooth": smooth.detach()}
    loss = img_loss + smooth_weight * smooth
    if feat_extractor is not None:
        b = R2.size(0)
        feat = feat_extractor(torch.cat((L2, R2, R1), dim=0))      # :104-105
        f_L2, f_R2, f_R1 = feat[:b], feat[b:2 * b], feat[2 * b:]  # :107
        feat_loss = LF.photometric_reconstruction_loss(f_R2, f_R1, f_L2, depth, T_2to1, batch["T_R2L"], batch["K"],
                                                       batch["Kinv"])                              # :109
        loss = img_loss + feat_weight * feat_loss + smooth_weight * smooth                          # :111
        terms["feat"] = feat_loss.detach()
    terms["total"] = loss.detach()
    return loss, terms


def depth_only_losses(depth_net, batch, img_scale=0.004, smooth_weight=10.0, depth_eps=1e-4):
    """BASELINE.json configs[0] (cfg 1): DispNetS alone with the stereo photometric loss (one view: the left image warped
    into the right one with the fixed baseline pose) + 10 * smooth -- unsupervise.py:94-102 without the pose network."""
    R2, L2 = batch["img_R2"], batch["img_L2"]
    depth = reciprocal(depth_net(R2)[0], depth_eps)                # [B,1,H,W]
    from dvf.ops import PhotoLossFn
    photo = PhotoLossFn.apply(R2, depth.squeeze(1), batch["T_R2L"].unsqueeze(0).contiguous(), batch["K"], batch["Kinv"], None,
                              (0, float(img_scale)), L2)
    smooth = LF.smooth_loss(depth)
    loss = photo + smooth_weight * smooth
    return loss, {"img": photo.detach(), "smooth": smooth.detach(), "total": loss.detach()}


def train_sfm_losses(disp_net, pose_exp_net, batch, w1=1.0, w2=0.0, w3=0.1, smooth_factor=2.0,
                     rotation_mode="euler", padding_mode="zeros", feat_extractor=None, feat_weight=0.1):
    """train.py:179-203.  feat_extractor: BASELINE.json configs[3] (cfg 4) adds the feature-reconstruction term of
    unsupervise.py:104-111 (single scale, finest depth, the network's two poses) to the 4-scale body."""
    tgt = batch["img_R2"]
    refs = [batch["img_R1"], batch["img_L2"]] + list(batch.get("extra_refs", []))[: pose_exp_net.nb_ref_imgs - 2]
    (masks, pose), disparities = _side_by_side(lambda: pose_exp_net(tgt, refs), lambda: disp_net(tgt))   # train.py:187,189
    depth = [reciprocal(d, 0.0) for d in disparities]              # :188
    l1 = LS.photometric_reconstruction_loss(tgt, refs, batch["K"], batch["Kinv"], depth, masks, pose,
                                            rotation_mode, padding_mode)                           # :191
    l3 = LS.smooth_loss(depth, smooth_factor)                      # :200
    l4 = torch.nn.functional.mse_loss(pose[:, 1], batch["T_R2L"])  # :201  (24 floats: left to torch)
    loss = w1 * l1 + w3 * l3 + l4
    terms = {"photo": l1.detach(), "smooth": l3.detach(), "lr": l4.detach()}
    if w2 > 0:
        l2 = LS.explainability_loss(masks)                         # :195-196
        loss = loss + w2 * l2
        terms["exp"] = l2.detach()
    if feat_extractor is not None:
        b = tgt.size(0)
        feat = feat_extractor(torch.cat((refs[1], tgt, refs[0]), dim=0))    # (L2, R2, R1) as unsupervise.py:104-105
        f_L2, f_R2, f_R1 = feat[:b], feat[b:2 * b], feat[2 * b:]
        lf = LF.photometric_reconstruction_loss(f_R2, f_R1, f_L2, depth[0].squeeze(1), pose[:, 0], pose[:, 1], batch["K"],
                                                batch["Kinv"], rotation_mode, padding_mode)
        loss = loss + feat_weight * lf
        terms["feat"] = lf.detach()
    terms["total"] = loss.detach()
    return loss, terms


def unsupervise_dvo_losses(depth_net, pose_net, batch, img_scale=0.004, smooth_weight=10.0, depth_eps=1e-4):
    """unsupervise_dvo.py:83-122 (the Caffe-style chain): T = (T_R2L, T_2to1) as se(3) vectors -> exponential map
    (t = R u) -> GeoTransform -> PinHole -> InverseWarping in PIXEL coordinates -> exact-zero-masked L1 for the
    stereo (L2 -> R2) and temporal (R1 -> R2) warps + 10 * smooth(depth).  All of it is the fused warp kernel with the
    DVF_POSE_SE3 | DVF_PIXEL_COORDS front end.  batch["T_R2L"] must be in the se(3) order (w, u), i.e. (0,0,0,Tx,0,0)
    as the reference dataset writes it (data/dataset_builder.py:155)."""
    from dvf.ops import PhotoLossFn
    R2, R1, L2 = batch["img_R2"], batch["img_R1"], batch["img_L2"]
    (_, T_2to1), disps = _side_by_side(lambda: pose_net((R2, R1)), lambda: depth_net(R2))
    inv_depth = disps[0]
    depth = reciprocal(inv_depth, depth_eps)                               # [B,1,H,W]
    pose = torch.stack((batch["T_R2L"], T_2to1), dim=0)                    # [V=2,B,6]: view 0 = left image, view 1 = R1
    photo = PhotoLossFn.apply(R2, depth.squeeze(1), pose, batch["K"], batch["Kinv"], None,
                              (L.POSE_SE3 | L.PIXEL_COORDS, float(img_scale)), L2, R1)
    smooth = LF.smooth_loss(depth)
    loss = photo + smooth_weight * smooth
    return loss, {"photo": photo.detach(), "smooth": smooth.detach(), "total": loss.detach()}


def paper_losses(depth_net, pose_net, batch, feat_extractor=None, img_scale=0.004, smooth_weight=10.0, feat_weight=0.1,
                 depth_eps=1e-4):
    """The loss of the reference's Caffe experiment ``depth_odometry_feature`` (train.prototxt:4355-4367, 4428-4446,
    4452-4661, 5571-5588): AbsLoss warp errors of the left and of the previous right image under the
    se(3) / pixel-coordinate chain (weight 1 each), edge-aware first-order smoothness of the inverse depth (weight 10)
    and, with ``feat_extractor`` (frozen: its parameters must not require gradients), 0.1 * AbsLoss on warped
    feature maps.  batch["T_R2L_se3"] is the stereo pose as the dataset holds it."""
    import loss_functions_caffe as LC
    R2, R1, L2 = batch["img_R2"], batch["img_R1"], batch["img_L2"]
    (_, T_2to1), disps = _side_by_side(lambda: pose_net((R2, R1)), lambda: depth_net(R2))
    inv_depth = disps[0]
    depth = reciprocal(inv_depth, depth_eps)                               # depth = (inv_depth + 1e-4)^-1  (:4355-4367)
    poses = (batch["T_R2L_se3"], T_2to1)
    photo = LC.abs_warp_loss(R2, (L2, R1), depth, poses, batch["K"], batch["Kinv"], img_scale=img_scale)
    smooth = LC.edge_aware_smooth_loss(inv_depth, R2, img_scale=img_scale)
    loss = photo + smooth_weight * smooth
    terms = {"photo": photo.detach(), "smooth": smooth.detach()}
    if feat_extractor is not None:
        b = R2.size(0)
        with torch.no_grad():                                              # frozen extractor: no graph, no weight gradients
            feat = feat_extractor(torch.cat((L2, R2, R1), dim=0))
        f_L2, f_R2, f_R1 = feat[:b], feat[b:2 * b], feat[2 * b:]
        lf = LC.abs_warp_loss(f_R2, (f_L2, f_R1), depth, poses, batch["K"], batch["Kinv"])
        loss = loss + feat_weight * lf
        terms["feat"] = lf.detach()
    terms["total"] = loss.detach()
    return loss, terms
