#!/usr/bin/env python3
"""Golden-vector generator.  Runs ONLY in the build container (it needs /root/reference).

Imports the reference's torch-only modules from /root/reference/pytorch_version (read-only,
PYTHONDONTWRITEBYTECODE), runs them on seeded synthetic inputs on the CPU and writes small
.npz fixtures (inputs + expected outputs + gradients) next to this file.  The fixtures are
DATA; no reference source text is stored.  The GPU box never runs this script.

    PYTHONDONTWRITEBYTECODE=1 python tests/golden/gen_golden.py
"""
import os
import sys
import warnings

import numpy as np
import torch

REF = "/root/reference/pytorch_version"
HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.dont_write_bytecode = True
sys.path.insert(0, ROOT)

from oracle import nets as onets          # noqa: E402  (build-side weight filler / layer tables only)
from oracle import steps as osteps        # noqa: E402  (build-side synthetic batch recipe only)


def _np(t):
    return t.detach().cpu().numpy()


def save(name, **arrs):
    path = os.path.join(HERE, name + ".npz")
    np.savez_compressed(path, **{k: np.asarray(v) for k, v in arrs.items()})
    print(f"{name}: {os.path.getsize(path)/1024:.1f} KiB")


def kitti_K(b, h, w, dtype):
    K = torch.tensor([[0.58 * w, 0, 0.5 * w], [0, 1.92 * h, 0.5 * h], [0, 0, 1.0]], dtype=torch.float64)
    return K.to(dtype).expand(b, 3, 3).contiguous(), torch.inverse(K).to(dtype).expand(b, 3, 3).contiguous()


def warp_inputs(seed, b, c, h, w, dtype, shift):
    g = torch.Generator().manual_seed(seed)
    img = torch.rand(b, c, h, w, generator=g, dtype=torch.float64).to(dtype)
    depth = (torch.rand(b, h, w, generator=g, dtype=torch.float64) * 29 + 1).to(dtype)
    pose = (torch.randn(b, 6, generator=g, dtype=torch.float64) * 0.05)
    pose[:, 0] += shift           # x translation so that >=10% of pixels leave the image
    pose = pose.to(dtype)
    wt = torch.randn(b, c, h, w, generator=g, dtype=torch.float64).to(dtype)
    K, Kinv = kitti_K(b, h, w, dtype)
    return img, depth, pose, K, Kinv, wt


def gen_warp():
    import inverse_warp as ref_iw
    import loss_functions as ref_lf
    for c, (h, w), dt, rot, pad in [
        (3, (16, 24), torch.float32, "euler", "zeros"),
        (3, (16, 24), torch.float32, "quat", "zeros"),
        (3, (16, 24), torch.float32, "euler", "border"),
        (3, (32, 104), torch.float32, "euler", "zeros"),
        (3, (16, 24), torch.float64, "euler", "zeros"),
        (32, (16, 24), torch.float32, "euler", "zeros"),
        (32, (24, 40), torch.float32, "quat", "border"),
    ]:
        img, depth, pose, K, Kinv, wt = warp_inputs(100 + c + h, 2, c, h, w, dt, 0.5)
        img.requires_grad_(True); depth.requires_grad_(True); pose.requires_grad_(True)
        fn = ref_iw.inverse_warp if c == 3 else ref_lf.inverse_warp      # the copy without the B3HW check
        ref_iw.pixel_coords = None     # module-global grid cache is keyed on height only and keeps its
        ref_lf.pixel_coords = None     # first dtype (inverse_warp.py:5,36-38): start every case clean
        out = fn(img, depth, pose, K, Kinv, rot, pad)
        (out * wt).sum().backward()
        tag = f"warp_c{c}_{h}x{w}_{'f32' if dt == torch.float32 else 'f64'}_{rot}_{pad}"
        frac_oob = float(((out == 0).all(1)).double().mean())
        save(tag, img=_np(img), depth=_np(depth), pose=_np(pose), K=_np(K), Kinv=_np(Kinv), wt=_np(wt),
             out=_np(out), g_img=_np(img.grad), g_depth=_np(depth.grad), g_pose=_np(pose.grad),
             frac_oob=frac_oob)


def gen_losses():
    import loss_functions as ref_lf
    import loss_functions_sfm as ref_sfm
    for c, (h, w) in [(3, (16, 24)), (3, (32, 104)), (32, (16, 24))]:
        b = 2
        g = torch.Generator().manual_seed(7 + c + h)
        mk = lambda: torch.rand(b, c, h, w, generator=g, dtype=torch.float64).float()
        R2, R1, L2 = mk(), mk(), mk()
        depth = (torch.rand(b, h, w, generator=g, dtype=torch.float64) * 29 + 1).float()
        T21 = (torch.randn(b, 6, generator=g, dtype=torch.float64) * 0.05).float()
        TRL = torch.tensor([-0.54, 0, 0, 0, 0, 0]).expand(b, 6).clone()
        TRL += (torch.randn(b, 6, generator=g, dtype=torch.float64) * 0.01).float()
        K, Kinv = kitti_K(b, h, w, torch.float32)
        leaves = [R2, R1, L2, depth, T21, TRL]
        for t in leaves:
            t.requires_grad_(True)
        loss = ref_lf.photometric_reconstruction_loss(R2, R1, L2, depth, T21, TRL, K, Kinv)
        loss.backward()
        save(f"photo_c{c}_{h}x{w}", R2=_np(R2), R1=_np(R1), L2=_np(L2), depth=_np(depth), T21=_np(T21),
             TRL=_np(TRL), K=_np(K), Kinv=_np(Kinv), loss=_np(loss), g_R2=_np(R2.grad), g_R1=_np(R1.grad),
             g_L2=_np(L2.grad), g_depth=_np(depth.grad), g_T21=_np(T21.grad), g_TRL=_np(TRL.grad))

    # smooth loss: single map (loss_functions, f=1 default) and 4-scale list (sfm, f=2)
    g = torch.Generator().manual_seed(11)
    maps = [(torch.rand(2, 1, 32 >> s, 52 >> s, generator=g, dtype=torch.float64) * 10 + 0.5).float().requires_grad_(True)
            for s in range(4)]
    l1 = ref_lf.smooth_loss(maps[0])
    l1.backward()
    g0 = maps[0].grad.clone(); maps[0].grad = None
    l4 = ref_sfm.smooth_loss(maps, 2.0)
    l4.backward()
    save("smooth", **{f"map{s}": _np(maps[s]) for s in range(4)}, loss_single=_np(l1), g_single=_np(g0),
         loss_multi=_np(l4), **{f"g_multi{s}": _np(maps[s].grad) for s in range(4)})

    # sfm 4-scale photometric with masks, 2 refs + explainability
    b, h, w = 2, 32, 64
    g = torch.Generator().manual_seed(13)
    tgt = (torch.rand(b, 3, h, w, generator=g, dtype=torch.float64) * 255).float()
    refs = [(torch.rand(b, 3, h, w, generator=g, dtype=torch.float64) * 255).float() for _ in range(2)]
    depth = [(torch.rand(b, 1, h >> s, w >> s, generator=g, dtype=torch.float64) * 29 + 1).float().requires_grad_(True)
             for s in range(4)]
    masks = [(torch.rand(b, 2, h >> s, w >> s, generator=g, dtype=torch.float64) * 0.9 + 0.05).float().requires_grad_(True)
             for s in range(4)]
    pose = (torch.randn(b, 2, 6, generator=g, dtype=torch.float64) * 0.03).float()
    pose[:, 1, 0] -= 0.54
    pose.requires_grad_(True)
    K, Kinv = kitti_K(b, h, w, torch.float32)
    for pad in ("zeros", "border"):
        for t in depth + masks + [pose]:
            t.grad = None
        l = ref_sfm.photometric_reconstruction_loss(tgt, refs, K, Kinv, depth, masks, pose, "euler", pad)
        l.backward()
        le = ref_sfm.explainability_loss(masks)
        gm_photo = [m.grad.clone() for m in masks]
        for m in masks:
            m.grad = None
        le.backward()
        save(f"photo_sfm_{pad}", tgt=_np(tgt), ref0=_np(refs[0]), ref1=_np(refs[1]), K=_np(K), Kinv=_np(Kinv),
             pose=_np(pose), loss=_np(l), exp_loss=_np(le), g_pose=_np(pose.grad),
             **{f"depth{s}": _np(depth[s]) for s in range(4)}, **{f"mask{s}": _np(masks[s]) for s in range(4)},
             **{f"g_depth{s}": _np(depth[s].grad) for s in range(4)},
             **{f"g_mask{s}": _np(gm_photo[s]) for s in range(4)},
             **{f"g_mask_exp{s}": _np(masks[s].grad) for s in range(4)})
    # no-mask variant (explainability_mask = None list)
    for t in depth + [pose]:
        t.grad = None
    l = ref_sfm.photometric_reconstruction_loss(tgt, refs, K, Kinv, depth, [None] * 4, pose, "euler", "zeros")
    l.backward()
    save("photo_sfm_nomask", loss=_np(l), g_pose=_np(pose.grad),
         **{f"g_depth{s}": _np(depth[s].grad) for s in range(4)})


def gen_stereo_pose():
    """The stereo pose AS THE REFERENCE'S DATASET HOLDS IT, (0, 0, 0, Tx, 0, 0) (data/dataset_builder.py:155), fed unchanged
    into loss_functions.photometric_reconstruction_loss -- which is what unsupervise.py:101 / train.py:201 do with real data
    (SURVEY preamble #6): pose_vec2mat reads it as zero translation and a rotation of Tx radians about x.  Pins the
    `--reference-stereo-pose` mode of the entry scripts (un_dataset.to_batch(reference_stereo_pose=True))."""
    import loss_functions as ref_lf
    b, c, h, w = 2, 3, 32, 104
    g = torch.Generator().manual_seed(77)
    mk = lambda: torch.rand(b, c, h, w, generator=g, dtype=torch.float64).float()
    R2, R1, L2 = mk(), mk(), mk()
    depth = (torch.rand(b, h, w, generator=g, dtype=torch.float64) * 29 + 1).float()
    T21 = (torch.randn(b, 6, generator=g, dtype=torch.float64) * 0.05).float()
    TRL = torch.tensor([0, 0, 0, -0.54, 0, 0], dtype=torch.float32).expand(b, 6).clone()
    K, Kinv = kitti_K(b, h, w, torch.float32)
    for t in (R2, R1, L2, depth, T21, TRL):
        t.requires_grad_(True)
    loss = ref_lf.photometric_reconstruction_loss(R2, R1, L2, depth, T21, TRL, K, Kinv)
    loss.backward()
    save("photo_c3_32x104_rawpose", R2=_np(R2), R1=_np(R1), L2=_np(L2), depth=_np(depth), T21=_np(T21),
         TRL=_np(TRL), K=_np(K), Kinv=_np(Kinv), loss=_np(loss), g_R2=_np(R2.grad), g_R1=_np(R1.grad),
         g_L2=_np(L2.grad), g_depth=_np(depth.grad), g_T21=_np(T21.grad), g_TRL=_np(TRL.grad))


def grad_digest(named_grads):
    """Per-parameter (l2 norm, sum, first 8 elements) in fp64: small but sensitive."""
    keys = sorted(named_grads)
    norms = np.array([float(named_grads[k].double().norm()) for k in keys])
    sums = np.array([float(named_grads[k].double().sum()) for k in keys])
    heads = np.stack([np.pad(_np(named_grads[k].double().flatten()[:8]), (0, max(0, 8 - named_grads[k].numel())))
                      for k in keys])
    return {"keys": np.array(keys), "norms": norms, "sums": sums, "heads": heads}


def gen_nets():
    import DispNetS as ref_disp
    import PoseExpNet as ref_pose
    import PoseExpNet_sfm as ref_pose_sfm
    import feat_extractor as ref_feat
    h, w = 48, 80          # not a multiple of 128: exercises crop_like (DispNetS.py:37-39)
    g = torch.Generator().manual_seed(21)
    x = (torch.rand(1, 3, h, w, generator=g, dtype=torch.float64) * 2 - 1).float()

    # --- DispNetS
    net = ref_disp.DispNetS()
    sd = onets.fill_params(onets.dispnet_layers(), seed=1)
    assert set(sd) == set(net.state_dict()) and all(sd[k].shape == v.shape for k, v in net.state_dict().items())
    net.load_state_dict(sd)
    net.train()
    outs = net(x)
    wts = [torch.randn(o.shape, generator=g, dtype=torch.float64).float() for o in outs]
    sum((o * t).sum() for o, t in zip(outs, wts)).backward()
    save("net_dispnet", x=_np(x), **{f"out{i}": _np(o) for i, o in enumerate(outs)},
         **{f"wt{i}": _np(t) for i, t in enumerate(wts)},
         **grad_digest({k: p.grad for k, p in net.named_parameters()}),
         n_params=sum(p.numel() for p in net.parameters()))

    # --- PoseExpNet_sfm (9-ch input, nb_ref=2) and PoseExpNet (6-ch input)
    for tag, net, cin, n_out, sfm in (("sfm", ref_pose_sfm.PoseExpNet(nb_ref_imgs=2, output_exp=True), 9, 12, True),
                                      ("six", ref_pose.PoseExpNet(output_exp=True), 6, 6, False)):
        sd = onets.fill_params(onets.posenet_layers(cin, n_out, 2, True), seed=2)
        assert set(sd) == set(net.state_dict())
        net.load_state_dict(sd)
        net.train()
        xin = (torch.rand(1, cin, h, w, generator=g, dtype=torch.float64) * 2 - 1).float()
        if sfm:
            masks, pose = net(xin[:, :3], [xin[:, 3:6], xin[:, 6:9]])
        else:
            masks, pose = net(xin)
        wp = torch.randn(pose.shape, generator=g, dtype=torch.float64).float()
        wm = [torch.randn(m.shape, generator=g, dtype=torch.float64).float() for m in masks]
        ((pose * wp).sum() * 100 + sum((m * t).sum() for m, t in zip(masks, wm))).backward()
        save(f"net_posenet_{tag}", x=_np(xin), pose=_np(pose), wp=_np(wp),
             **{f"mask{i}": _np(m) for i, m in enumerate(masks)}, **{f"wm{i}": _np(t) for i, t in enumerate(wm)},
             **grad_digest({k: p.grad for k, p in net.named_parameters()}))

    # --- FeatExtractor
    net = ref_feat.FeatExtractor()
    sd = onets.fill_params(onets.featnet_layers(), seed=3)
    assert set(sd) == set(net.state_dict()) and all(sd[k].shape == v.shape for k, v in net.state_dict().items())
    net.load_state_dict(sd)
    xf = (torch.rand(2, 3, 32, 64, generator=g, dtype=torch.float64) * 2 - 1).float()
    out = net(xf)
    wt = torch.randn(out.shape, generator=g, dtype=torch.float64).float()
    (out * wt).sum().backward()
    save("net_featnet", x=_np(xf), out=_np(out), wt=_np(wt),
         **grad_digest({k: p.grad for k, p in net.named_parameters()}))


def param_digest(sd, sd0=None):
    """Per-parameter sum and norm; with the initial values `sd0` also sum and norm of the UPDATE (p - p0): after two
    Adam steps every element has moved by ~2*lr, so these two bind where the parameter's own norm barely changes."""
    keys = sorted(sd)
    out = [np.array(keys), np.array([float(sd[k].double().sum()) for k in keys]),
           np.array([float(sd[k].double().norm()) for k in keys])]
    if sd0 is not None:
        out += [np.array([float((sd[k].double() - sd0[k].double()).sum()) for k in keys]),
                np.array([float((sd[k].double() - sd0[k].double()).norm()) for k in keys])]
    return out


def gen_steps():
    """End-to-end iterations with the REFERENCE modules, losses and torch.optim.Adam, following the
    loop bodies of train.py:179-214 and unsupervise.py:83-120 (the scripts themselves cannot be
    imported: argv parsing / missing deps / hard-coded dataset paths)."""
    import DispNetS as ref_disp
    import PoseExpNet as ref_pose
    import PoseExpNet_sfm as ref_pose_sfm
    import feat_extractor as ref_feat
    import loss_functions as ref_lf
    import loss_functions_sfm as ref_sfm
    import torch.nn.functional as F
    b, h, w = 2, 64, 128

    # ---- unsupervise-style (cfg 2 family; with features = cfg 3 family)
    for with_feat in (False, True):
        batch = osteps.synthetic_batch(b, h, w, seed=1234)
        disp_net, pose_net = ref_disp.DispNetS(), ref_pose.PoseExpNet(output_exp=True)
        disp_net.load_state_dict(onets.fill_params(onets.dispnet_layers(), seed=1))
        pose_net.load_state_dict(onets.fill_params(onets.posenet_layers(6, 6, 2, True), seed=2))
        groups = [{"params": pose_net.parameters(), "lr": 1e-3}, {"params": disp_net.parameters(), "lr": 1e-3}]
        feat_net = None
        if with_feat:
            feat_net = ref_feat.FeatExtractor()
            feat_net.load_state_dict(onets.fill_params(onets.featnet_layers(), seed=3))
            groups.append({"params": feat_net.parameters(), "lr": 1e-3})
        opt = torch.optim.Adam(groups, betas=(0.9, 0.999), eps=1e-8, weight_decay=1e-8)   # unsupervise.py:241
        rec = {}
        for it in range(2):
            R2, R1, L2 = batch["img_R2"], batch["img_R1"], batch["img_L2"]
            inv_depth = disp_net(R2)[0]
            _, T21 = pose_net(torch.cat((R2, R1), 1))
            depth = (1 / (inv_depth + 1e-4)).squeeze(1)
            img_l = ref_lf.photometric_reconstruction_loss(0.004 * R2, 0.004 * R1, 0.004 * L2, depth, T21,
                                                           batch["T_R2L"], batch["K"], batch["Kinv"])
            sm_l = ref_lf.smooth_loss(depth.unsqueeze(1))
            loss = img_l + 10 * sm_l
            if with_feat:
                feat = feat_net(torch.cat((L2, R2, R1), 0))
                f_L2, f_R2, f_R1 = feat[:b], feat[b:2 * b], feat[2 * b:]
                f_l = ref_lf.photometric_reconstruction_loss(f_R2, f_R1, f_L2, depth, T21, batch["T_R2L"],
                                                             batch["K"], batch["Kinv"])
                loss = img_l + 0.1 * f_l + 10 * sm_l
                rec[f"feat{it}"] = _np(f_l)
            opt.zero_grad()
            loss.backward()
            if it == 0:
                rec.update({"g_disp_" + k: v for k, v in grad_digest(
                    {k: p.grad for k, p in disp_net.named_parameters() if p.grad is not None}).items()})
                rec.update({"g_pose_" + k: v for k, v in grad_digest(
                    {k: p.grad for k, p in pose_net.named_parameters() if p.grad is not None}).items()})
                rec["T21_0"] = _np(T21)
            opt.step()
            rec[f"img{it}"], rec[f"smooth{it}"], rec[f"total{it}"] = _np(img_l), _np(sm_l), _np(loss)
        init = {"disp": onets.fill_params(onets.dispnet_layers(), seed=1), "pose": onets.fill_params(onets.posenet_layers(6, 6, 2, True), seed=2),
                "feat": onets.fill_params(onets.featnet_layers(), seed=3)}
        for name, net in (("disp", disp_net), ("pose", pose_net)) + ((("feat", feat_net),) if with_feat else ()):
            k, s, n, ds, dn = param_digest(net.state_dict(), init[name])
            rec[f"p_{name}_keys"], rec[f"p_{name}_sums"], rec[f"p_{name}_norms"] = k, s, n
            rec[f"p_{name}_dsums"], rec[f"p_{name}_dnorms"] = ds, dn
            for key, val in net.state_dict().items():         # small tensors in full: compared element by element
                if val.numel() <= 4096:
                    rec[f"p_{name}_val_{key}"] = _np(val)
        save("step_unsup_feat" if with_feat else "step_unsup", **rec, b=b, h=h, w=w)

    # ---- train.py-style (sfm; cfg 4/5 family): the base case, with the explainability term (w2 > 0, train.py:195),
    # and with nb_ref_imgs = 4 (cfg 5: five-frame window -> 15-channel pose network input, V = 4 warps per scale)
    for tag, nb_ref, w2, bb in (("step_train_sfm", 2, 0.0, b), ("step_train_sfm_exp", 2, 0.2, b), ("step_train_sfm_v4", 4, 0.2, 1)):
        batch = osteps.synthetic_batch(bb, h, w, seed=1234, n_views=nb_ref)
        disp_net, pose_net = ref_disp.DispNetS(), ref_pose_sfm.PoseExpNet(nb_ref_imgs=nb_ref, output_exp=True)
        disp_net.load_state_dict(onets.fill_params(onets.dispnet_layers(), seed=1))
        pose_net.load_state_dict(onets.fill_params(onets.posenet_layers(3 * (1 + nb_ref), 6 * nb_ref, nb_ref, True), seed=2))
        disp_net.train(); pose_net.train()
        opt = torch.optim.Adam([{"params": disp_net.parameters(), "lr": 2e-4},
                                {"params": pose_net.parameters(), "lr": 2e-4}], betas=(0.9, 0.999), weight_decay=0)
        rec = {}
        for it in range(2):
            tgt, refs = batch["img_R2"], [batch["img_R1"], batch["img_L2"]] + list(batch["extra_refs"])[:nb_ref - 2]
            disps = disp_net(tgt)
            depth = [1 / d for d in disps]
            masks, pose = pose_net(tgt, refs)
            l1 = ref_sfm.photometric_reconstruction_loss(tgt, refs, batch["K"], batch["Kinv"], depth, masks, pose,
                                                         "euler", "zeros")
            l2 = ref_sfm.explainability_loss(masks) if w2 > 0 else 0
            l3 = ref_sfm.smooth_loss(depth, 2.0)
            l4 = F.mse_loss(pose[:, 1], batch["T_R2L"])
            loss = 1.0 * l1 + w2 * l2 + 0.1 * l3 + l4
            opt.zero_grad()
            loss.backward()
            if it == 0:
                rec.update({"g_disp_" + k: v for k, v in grad_digest(
                    {k: p.grad for k, p in disp_net.named_parameters() if p.grad is not None}).items()})
                rec.update({"g_pose_" + k: v for k, v in grad_digest(
                    {k: p.grad for k, p in pose_net.named_parameters() if p.grad is not None}).items()})
                rec["pose_0"] = _np(pose)
            opt.step()
            rec[f"photo{it}"], rec[f"smooth{it}"], rec[f"lr{it}"], rec[f"total{it}"] = _np(l1), _np(l3), _np(l4), _np(loss)
            if w2 > 0:
                rec[f"exp{it}"] = _np(l2)
        init = {"disp": onets.fill_params(onets.dispnet_layers(), seed=1),
                "pose": onets.fill_params(onets.posenet_layers(3 * (1 + nb_ref), 6 * nb_ref, nb_ref, True), seed=2)}
        for name, net in (("disp", disp_net), ("pose", pose_net)):
            k, s, n, ds, dn = param_digest(net.state_dict(), init[name])
            rec[f"p_{name}_keys"], rec[f"p_{name}_sums"], rec[f"p_{name}_norms"] = k, s, n
            rec[f"p_{name}_dsums"], rec[f"p_{name}_dnorms"] = ds, dn
            for key, val in net.state_dict().items():         # small tensors in full: compared element by element
                if val.numel() <= 4096:
                    rec[f"p_{name}_val_{key}"] = _np(val)
        save(tag, **rec, b=bb, h=h, w=w, nb_ref=nb_ref, w2=w2)


def gen_metrics():
    """Depth metrics of the reference (loss_functions_sfm.compute_errors, :80-116): KITTI-like sparse ground truth
    (0 = no LiDAR return, some values beyond 80 m) against a noisy, differently scaled prediction."""
    import loss_functions_sfm as ref_sfm
    g = torch.Generator().manual_seed(41)
    n, hh, ww = 3, 96, 320
    gt = torch.rand(n, hh, ww, generator=g, dtype=torch.float64) * 95.0
    gt = torch.where(torch.rand(n, hh, ww, generator=g, dtype=torch.float64) < 0.7, torch.zeros_like(gt), gt).float()
    pred = (gt.double() * 0.37 * (1 + 0.25 * torch.randn(n, hh, ww, generator=g, dtype=torch.float64)) +
            torch.rand(n, hh, ww, generator=g, dtype=torch.float64) * 0.5).float()
    pred[0, :4] = -1.0                      # exercises the clamp(1e-3, 80)
    out = {}
    for crop in (True, False):
        out["crop" if crop else "full"] = np.array(ref_sfm.compute_errors(gt, pred, crop=crop), dtype=np.float64)
    save("metrics_compute_errors", gt=_np(gt), pred=_np(pred), **out)


def gen_se3():
    """SE3 exponential map forward/backward from the reference's se3_generate.py (numpy on CPU; its forward and
    backward end in .cuda(), which is an identity in this CPU-only generator process)."""
    torch.Tensor.cuda = lambda self, *a, **k: self
    import se3_generate as ref_se3
    g = torch.Generator().manual_seed(31)
    vec = torch.randn(6, 6, generator=g, dtype=torch.float64) * 0.3
    vec[0, :3] = 0.0                      # theta = 0: the first-order branch
    vec[1, :3] *= 1e-8
    x = vec.float().view(6, 6, 1, 1).requires_grad_(True)
    out = ref_se3.generate_se3(x)         # [6,1,4,4] float64
    wt = torch.randn(6, 1, 4, 4, generator=g, dtype=torch.float64)
    (out * wt).sum().backward()
    save("se3_expmap", vec=_np(vec.float()), out=_np(out), wt=_np(wt), g_vec=_np(x.grad.view(6, 6)))


def main():
    if not os.path.isdir(REF):
        sys.exit("reference not present: golden vectors can only be generated in the build container")
    sys.path.insert(0, REF)
    torch.set_num_threads(8)
    torch.manual_seed(0)
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")      # grid_sample align_corners default-change warning
        groups = {"warp": gen_warp, "losses": gen_losses, "nets": gen_nets, "steps": gen_steps, "metrics": gen_metrics,
                  "se3": gen_se3, "stereo": gen_stereo_pose}
        only = [a for a in sys.argv[1:] if a in groups]
        for name, fn in groups.items():          # usage: gen_golden.py [group ...]   (default: all)
            if not only or name in only:
                fn()


if __name__ == "__main__":
    main()
