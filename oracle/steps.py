"""Oracle (test infrastructure): Adam and the training-step bodies.

  adam_step          torch.optim.Adam as configured at train.py:150-156 / unsupervise.py:241
                     (L2-in-gradient weight decay, bias correction, eps outside the sqrt)
  step_unsupervise   unsupervise.py:83-120 with PoseExpNet substituted for FixOdometryNet
                     (north_star; FixOdometryNet is the out-of-scope fixed-point stack) and
                     disp[0] used where unsupervise.py:99 would break on DispNetS' list output
  step_train_sfm     train.py:179-214 (DispNetS + PoseExpNet_sfm, 4-scale photometric with masks,
                     smooth, stereo-pose MSE)

The same seed recipe (``synthetic_batch``) is used by the golden generator, the tests, the
product's entry scripts and ``bench.py``.
"""
import torch
import torch.nn.functional as F

from . import losses, nets


def synthetic_batch(b, h, w, seed=1234, rank=0, n_views=2, smooth_images=True, dtype=torch.float32):
    """Seeded KITTI-shaped inputs: images U[0,255), K=[[.58W,0,.5W],[0,1.92H,.5H],[0,0,1]],
    stereo pose (-0.54,0,0,0,0,0) in the (t, r) convention of pose_vec2mat."""
    g = torch.Generator().manual_seed(seed + rank)
    imgs = []
    for _ in range(1 + n_views):
        im = torch.rand(b, 3, h, w, generator=g, dtype=torch.float64) * 255.0
        if smooth_images:
            # 5x5 box low-pass (reflect) so that photometric gradients are meaningful
            im = F.avg_pool2d(F.pad(im, (2, 2, 2, 2), mode="reflect"), 5, stride=1)
        imgs.append(im.to(dtype))
    K = torch.tensor([[0.58 * w, 0, 0.5 * w], [0, 1.92 * h, 0.5 * h], [0, 0, 1.0]], dtype=torch.float64)
    Kinv = torch.inverse(K)
    K = K.to(dtype).expand(b, 3, 3).contiguous()
    Kinv = Kinv.to(dtype).expand(b, 3, 3).contiguous()
    T_R2L = torch.tensor([-0.54, 0, 0, 0, 0, 0], dtype=dtype).expand(b, 6).contiguous()
    # order follows the dataset tuple (img_R1, img_L2, img_R2, ...)  un_dataset.py:78-84
    return {"img_R2": imgs[0], "img_R1": imgs[1], "img_L2": imgs[2] if n_views >= 2 else imgs[1],
            "extra_refs": imgs[3:], "K": K, "Kinv": Kinv, "T_R2L": T_R2L}


def adam_init(params):
    return {"step": 0, "m": {k: torch.zeros_like(v) for k, v in params.items()},
            "v": {k: torch.zeros_like(v) for k, v in params.items()}}


def adam_step(params, grads, state, lr, beta1=0.9, beta2=0.999, eps=1e-8, weight_decay=0.0):
    """In-place single-tensor Adam (torch/optim/adam.py _single_tensor_adam, amsgrad off)."""
    state["step"] += 1
    t = state["step"]
    bc1 = 1 - beta1 ** t
    bc2 = 1 - beta2 ** t
    for k, p in params.items():
        g = grads[k]
        if weight_decay != 0:
            g = g + weight_decay * p
        m, v = state["m"][k], state["v"][k]
        m.mul_(beta1).add_(g, alpha=1 - beta1)
        v.mul_(beta2).addcmul_(g, g, value=1 - beta2)
        denom = (v.sqrt() / (bc2 ** 0.5)).add_(eps)
        p.addcdiv_(m, denom, value=-lr / bc1)


def _leaf(sd):
    return {k: v.detach().clone().requires_grad_(True) for k, v in sd.items()}


def step_unsupervise(disp_sd, pose_sd, batch, adam_state=None, lr=1e-3, weight_decay=1e-8,
                     img_scale=0.004, smooth_weight=10.0, feat_sd=None, feat_weight=0.1,
                     do_update=True, dvo=False):
    """One unsupervise.py iteration.  Returns dict(losses, grads) and updates the dicts in place.  dvo=True swaps the
    image term for the unsupervise_dvo.py chain (se3 exponential map + pixel-coordinate warp, batch["T_R2L"] then in
    (w, u) order)."""
    dsd, psd = _leaf(disp_sd), _leaf(pose_sd)
    fsd = _leaf(feat_sd) if feat_sd is not None else None
    R2, R1, L2 = batch["img_R2"], batch["img_R1"], batch["img_L2"]
    disp = nets.dispnet_forward(dsd, R2)[0]                                   # unsupervise.py:94
    _, T_2to1 = nets.posenet_forward(psd, torch.cat((R2, R1), 1), 2, True, sfm=False)   # :92,95
    depth = (1 / (disp + 1e-4)).squeeze(1)                                    # :99
    if dvo:
        from . import geometry
        img_loss = geometry.dvo_photometric_loss(img_scale * R2, img_scale * L2, img_scale * R1, depth,
                                                 batch["T_R2L"], T_2to1, batch["K"])   # unsupervise_dvo.py:96-117
    else:
        img_loss = losses.photometric_reconstruction_loss(img_scale * R2, img_scale * R1, img_scale * L2,
                                                          depth, T_2to1, batch["T_R2L"], batch["K"],
                                                          batch["Kinv"])      # :101
    smooth = losses.smooth_loss(depth.unsqueeze(1))                           # :102
    total = img_loss + smooth_weight * smooth
    out = {"img": img_loss.detach(), "smooth": smooth.detach()}
    if fsd is not None:
        b = R2.shape[0]
        feat = nets.featnet_forward(fsd, torch.cat((L2, R2, R1), 0))          # :104-105
        f_L2, f_R2, f_R1 = feat[:b], feat[b:2 * b], feat[2 * b:]              # :107
        feat_loss = losses.photometric_reconstruction_loss(f_R2, f_R1, f_L2, depth, T_2to1,
                                                           batch["T_R2L"], batch["K"], batch["Kinv"])  # :109
        total = total + feat_weight * feat_loss                               # :111
        out["feat"] = feat_loss.detach()
    out["total"] = total.detach()
    total.backward()
    groups = [("pose", pose_sd, psd), ("disp", disp_sd, dsd)] + ([("feat", feat_sd, fsd)] if fsd is not None else [])
    grads = {g: {k: v.grad for k, v in leaf.items() if v.grad is not None} for g, _, leaf in groups}
    if do_update:
        if adam_state is None:
            adam_state = {}
        for g, sd, leaf in groups:
            st = adam_state.setdefault(g, adam_init(sd))
            zero = {k: torch.zeros_like(v) for k, v in sd.items()}
            with torch.no_grad():
                # parameters without a gradient are skipped by torch.optim (p.grad is None)
                upd = {k: v for k, v in sd.items() if k in grads[g]}
                sub = {"step": st["step"], "m": {k: st["m"][k] for k in upd}, "v": {k: st["v"][k] for k in upd}}
                adam_step(upd, grads[g], sub, lr, weight_decay=weight_decay)
                st["step"] = sub["step"]
            del zero
    return out, grads, adam_state


def step_depth_only(disp_sd, batch, adam_state=None, lr=1e-3, weight_decay=1e-8, img_scale=0.004, smooth_weight=10.0,
                    do_update=True):
    """BASELINE.json configs[0] (cfg 1): DispNetS alone, stereo photometric loss (the left image warped into the right
    one with the fixed baseline pose, one view) + 10 * smooth -- the unsupervise.py body (:94-102) without the pose
    network and the temporal view; the loss is loss_functions_sfm's single-scale, single-view form."""
    dsd = _leaf(disp_sd)
    R2, L2 = batch["img_R2"], batch["img_L2"]
    disp = nets.dispnet_forward(dsd, R2)[0]
    depth = 1 / (disp + 1e-4)
    photo = losses.photometric_reconstruction_loss_sfm(img_scale * R2, [img_scale * L2], batch["K"], batch["Kinv"], [depth],
                                                       [None], batch["T_R2L"].unsqueeze(1))
    smooth = losses.smooth_loss(depth)
    total = photo + smooth_weight * smooth
    out = {"img": photo.detach(), "smooth": smooth.detach(), "total": total.detach()}
    total.backward()
    grads = {"disp": {k: v.grad for k, v in dsd.items() if v.grad is not None}}
    if do_update:
        if adam_state is None:
            adam_state = {}
        st = adam_state.setdefault("disp", adam_init(disp_sd))
        with torch.no_grad():
            upd = {k: v for k, v in disp_sd.items() if k in grads["disp"]}
            sub = {"step": st["step"], "m": {k: st["m"][k] for k in upd}, "v": {k: st["v"][k] for k in upd}}
            adam_step(upd, grads["disp"], sub, lr, weight_decay=weight_decay)
            st["step"] = sub["step"]
    return out, grads, adam_state


def loss_only(batch, depth, T_2to1, img_scale=0.004, smooth_weight=10.0):
    """BASELINE.md section 3's loss-only micro-benchmark body: 2 image warps (C=3) + smooth, forward + backward to
    depth and pose."""
    depth = depth.detach().clone().requires_grad_(True)
    T = T_2to1.detach().clone().requires_grad_(True)
    l = losses.photometric_reconstruction_loss(img_scale * batch["img_R2"], img_scale * batch["img_R1"], img_scale * batch["img_L2"],
                                               depth, T, batch["T_R2L"], batch["K"], batch["Kinv"]) + \
        smooth_weight * losses.smooth_loss(depth.unsqueeze(1))
    l.backward()
    return l.detach(), depth.grad, T.grad


def step_train_sfm(disp_sd, pose_sd, batch, adam_state=None, lr=2e-4, w1=1.0, w2=0.0, w3=0.1,
                   smooth_factor=2.0, nb_ref_imgs=2, rotation_mode="euler", padding_mode="zeros",
                   do_update=True, feat_sd=None, feat_weight=0.1):
    """One train.py iteration (train.py:179-214).  feat_sd: BASELINE.json configs[3] (cfg 4) adds the feature
    reconstruction term of unsupervise.py:104-111 (FeatExtractor on the three frames, single-scale photometric loss on
    the 32-channel maps with the finest depth and the network's two poses, weight 0.1) to the 4-scale body."""
    dsd, psd = _leaf(disp_sd), _leaf(pose_sd)
    fsd = _leaf(feat_sd) if feat_sd is not None else None
    tgt = batch["img_R2"]
    refs = [batch["img_R1"], batch["img_L2"]] + list(batch.get("extra_refs", []))[: nb_ref_imgs - 2]
    disps = nets.dispnet_forward(dsd, tgt)                                    # :187
    depth = [1 / d for d in disps]                                            # :188
    masks, pose = nets.posenet_forward(psd, torch.cat([tgt] + refs, 1), nb_ref_imgs, True, sfm=True)  # :189
    l1 = losses.photometric_reconstruction_loss_sfm(tgt, refs, batch["K"], batch["Kinv"], depth, masks,
                                                    pose, rotation_mode, padding_mode)   # :191
    l2 = losses.explainability_loss(masks) if w2 > 0 else 0                   # :195-198
    l3 = losses.smooth_loss(depth, smooth_factor)                             # :200
    l4 = F.mse_loss(pose[:, 1], batch["T_R2L"])                               # :201
    total = w1 * l1 + w2 * l2 + w3 * l3 + l4                                  # :203
    out = {"photo": l1.detach(), "smooth": l3.detach(), "lr": l4.detach()}
    if w2 > 0:
        out["exp"] = l2.detach()
    if fsd is not None:
        b = tgt.shape[0]
        feat = nets.featnet_forward(fsd, torch.cat((refs[1], tgt, refs[0]), 0))          # (L2, R2, R1) as unsupervise.py:104
        f_L2, f_R2, f_R1 = feat[:b], feat[b:2 * b], feat[2 * b:]
        lf = losses.photometric_reconstruction_loss(f_R2, f_R1, f_L2, depth[0].squeeze(1), pose[:, 0], pose[:, 1],
                                                    batch["K"], batch["Kinv"], rotation_mode, padding_mode)
        total = total + feat_weight * lf
        out["feat"] = lf.detach()
    out["total"] = total.detach()
    total.backward()
    groups = [("disp", disp_sd, dsd), ("pose", pose_sd, psd)] + ([("feat", feat_sd, fsd)] if fsd is not None else [])
    grads = {g: {k: v.grad for k, v in leaf.items() if v.grad is not None} for g, _, leaf in groups}
    if do_update:
        if adam_state is None:
            adam_state = {}
        for g, sd, leaf in groups:
            st = adam_state.setdefault(g, adam_init(sd))
            with torch.no_grad():
                upd = {k: v for k, v in sd.items() if k in grads[g]}
                sub = {"step": st["step"], "m": {k: st["m"][k] for k in upd}, "v": {k: st["v"][k] for k in upd}}
                adam_step(upd, grads[g], sub, lr, weight_decay=0.0)
                st["step"] = sub["step"]
    return out, grads, adam_state


def step_paper(disp_sd, pose_sd, batch, feat_sd=None, img_scale=0.004, smooth_weight=10.0, feat_weight=0.1):
    """The Caffe experiment's loss (depth_odometry_feature/train.prototxt) on the PyTorch networks:
    AbsLoss warp errors (se3 / pixel-coordinate chain) + 10 * edge-aware smoothness (+ 0.1 * feature AbsLoss with a
    frozen extractor).  Gradients only (no update); batch["T_R2L_se3"] in (w, u) order."""
    from . import geometry
    dsd, psd = _leaf(disp_sd), _leaf(pose_sd)
    R2, R1, L2 = batch["img_R2"], batch["img_R1"], batch["img_L2"]
    inv_depth = nets.dispnet_forward(dsd, R2)[0]
    _, T_2to1 = nets.posenet_forward(psd, torch.cat((R2, R1), 1), 2, True, sfm=False)
    depth = (1 / (inv_depth + 1e-4)).squeeze(1)
    photo = geometry.dvo_photometric_loss(img_scale * R2, img_scale * L2, img_scale * R1, depth, batch["T_R2L_se3"], T_2to1,
                                          batch["K"], caffe_abs=True)
    smooth = losses.edge_aware_smooth_caffe(inv_depth, img_scale * R2)
    total = photo + smooth_weight * smooth
    out = {"photo": photo.detach(), "smooth": smooth.detach()}
    if feat_sd is not None:
        b = R2.shape[0]
        with torch.no_grad():
            feat = nets.featnet_forward(feat_sd, torch.cat((L2, R2, R1), 0))
        f_L2, f_R2, f_R1 = feat[:b], feat[b:2 * b], feat[2 * b:]
        lf = geometry.dvo_photometric_loss(f_R2, f_L2, f_R1, depth, batch["T_R2L_se3"], T_2to1, batch["K"], caffe_abs=True)
        total = total + feat_weight * lf
        out["feat"] = lf.detach()
    out["total"] = total.detach()
    total.backward()
    grads = {"disp": {k: v.grad for k, v in dsd.items() if v.grad is not None},
             "pose": {k: v.grad for k, v in psd.items() if v.grad is not None}}
    return out, grads
