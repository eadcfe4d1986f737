"""Pins the CPU oracle (oracle/) against golden vectors produced by the reference itself
(tests/golden/gen_golden.py).  CPU only.  Bound: 1e-4 relative (max-abs error over max-abs
reference), the tolerance north_star states for fp32; most cases are orders below it."""
import numpy as np
import pytest
import torch

from conftest import load_golden, rel_err, t
from oracle import geometry, losses, nets, steps

TOL = 1e-4


@pytest.mark.parametrize("name", [
    "warp_c3_16x24_f32_euler_zeros", "warp_c3_16x24_f32_quat_zeros", "warp_c3_16x24_f32_euler_border",
    "warp_c3_32x104_f32_euler_zeros", "warp_c3_16x24_f64_euler_zeros", "warp_c32_16x24_f32_euler_zeros",
    "warp_c32_24x40_f32_quat_border"])
def test_inverse_warp(name):
    g = load_golden(name)
    rot = "quat" if "quat" in name else "euler"
    pad = "border" if "border" in name else "zeros"
    img, depth, pose = (t(g[k]).requires_grad_(True) for k in ("img", "depth", "pose"))
    out = geometry.inverse_warp(img, depth, pose, t(g["K"]), t(g["Kinv"]), rot, pad)
    (out * t(g["wt"])).sum().backward()
    assert rel_err(out, g["out"]) < TOL
    assert rel_err(img.grad, g["g_img"]) < TOL
    assert rel_err(depth.grad, g["g_depth"]) < TOL
    assert rel_err(pose.grad, g["g_pose"]) < TOL
    if pad == "zeros":
        assert float(g["frac_oob"]) >= 0.10        # the fixtures do exercise the exact-zero mask
        assert np.array_equal((out.detach().numpy() == 0).all(1), (g["out"] == 0).all(1))


@pytest.mark.parametrize("name", ["photo_c3_16x24", "photo_c3_32x104", "photo_c32_16x24", "photo_c3_32x104_rawpose"])
def test_photometric_single_scale(name):
    g = load_golden(name)
    leaves = {k: t(g[k]).requires_grad_(True) for k in ("R2", "R1", "L2", "depth", "T21", "TRL")}
    loss = losses.photometric_reconstruction_loss(leaves["R2"], leaves["R1"], leaves["L2"], leaves["depth"],
                                                  leaves["T21"], leaves["TRL"], t(g["K"]), t(g["Kinv"]))
    loss.backward()
    assert rel_err(loss, g["loss"]) < TOL
    for k in leaves:
        assert rel_err(leaves[k].grad, g["g_" + k]) < TOL, k


def test_smooth_loss():
    g = load_golden("smooth")
    maps = [t(g[f"map{s}"]).requires_grad_(True) for s in range(4)]
    l1 = losses.smooth_loss(maps[0])
    l1.backward()
    assert rel_err(l1, g["loss_single"]) < 1e-6
    assert rel_err(maps[0].grad, g["g_single"]) < 1e-6
    maps[0].grad = None
    l4 = losses.smooth_loss(maps, 2.0)
    l4.backward()
    assert rel_err(l4, g["loss_multi"]) < 1e-6
    for s in range(4):
        assert rel_err(maps[s].grad, g[f"g_multi{s}"]) < 1e-6


@pytest.mark.parametrize("pad", ["zeros", "border"])
def test_photometric_sfm(pad):
    g = load_golden(f"photo_sfm_{pad}")
    depth = [t(g[f"depth{s}"]).requires_grad_(True) for s in range(4)]
    masks = [t(g[f"mask{s}"]).requires_grad_(True) for s in range(4)]
    pose = t(g["pose"]).requires_grad_(True)
    refs = [t(g["ref0"]), t(g["ref1"])]
    loss = losses.photometric_reconstruction_loss_sfm(t(g["tgt"]), refs, t(g["K"]), t(g["Kinv"]), depth, masks,
                                                      pose, "euler", pad)
    loss.backward()
    assert rel_err(loss, g["loss"]) < TOL
    assert rel_err(pose.grad, g["g_pose"]) < TOL
    for s in range(4):
        assert rel_err(depth[s].grad, g[f"g_depth{s}"]) < TOL
        assert rel_err(masks[s].grad, g[f"g_mask{s}"]) < TOL
    for m in masks:
        m.grad = None
    le = losses.explainability_loss(masks)
    le.backward()
    assert rel_err(le, g["exp_loss"]) < 1e-6
    for s in range(4):
        assert rel_err(masks[s].grad, g[f"g_mask_exp{s}"]) < 1e-6
    if pad == "zeros":
        gn = load_golden("photo_sfm_nomask")
        for x in depth + [pose]:
            x.grad = None
        l2 = losses.photometric_reconstruction_loss_sfm(t(g["tgt"]), refs, t(g["K"]), t(g["Kinv"]), depth,
                                                        [None] * 4, pose)
        l2.backward()
        assert rel_err(l2, gn["loss"]) < TOL
        assert rel_err(pose.grad, gn["g_pose"]) < TOL
        for s in range(4):
            assert rel_err(depth[s].grad, gn[f"g_depth{s}"]) < TOL


def _check_digest(g, grads, prefix=""):
    keys = [str(k) for k in g[prefix + "keys"]]
    assert sorted(grads) == keys
    for i, k in enumerate(keys):
        gr = grads[k].double()
        scale = max(float(g[prefix + "norms"][i]), 1e-12)
        assert abs(float(gr.norm()) - float(g[prefix + "norms"][i])) / scale < TOL, k
        n = min(8, gr.numel())
        assert np.abs(gr.flatten()[:n].numpy() - g[prefix + "heads"][i][:n]).max() / scale < TOL, k


def test_dispnet():
    g = load_golden("net_dispnet")
    sd = {k: v.requires_grad_(True) for k, v in nets.fill_params(nets.dispnet_layers(), seed=1).items()}
    assert sum(v.numel() for v in sd.values()) == int(g["n_params"]) == 31596900
    outs = nets.dispnet_forward(sd, t(g["x"]))
    for i, o in enumerate(outs):
        assert rel_err(o, g[f"out{i}"]) < TOL
    sum((o * t(g[f"wt{i}"])).sum() for i, o in enumerate(outs)).backward()
    _check_digest(g, {k: v.grad for k, v in sd.items()})


@pytest.mark.parametrize("tag,cin,nout,sfm", [("sfm", 9, 12, True), ("six", 6, 6, False)])
def test_posenet(tag, cin, nout, sfm):
    g = load_golden(f"net_posenet_{tag}")
    sd = {k: v.requires_grad_(True) for k, v in nets.fill_params(nets.posenet_layers(cin, nout, 2, True), seed=2).items()}
    masks, pose = nets.posenet_forward(sd, t(g["x"]), 2, True, sfm=sfm)
    assert rel_err(pose, g["pose"]) < TOL
    for i, m in enumerate(masks):
        assert rel_err(m, g[f"mask{i}"]) < TOL
    ((pose * t(g["wp"])).sum() * 100 + sum((m * t(g[f"wm{i}"])).sum() for i, m in enumerate(masks))).backward()
    _check_digest(g, {k: v.grad for k, v in sd.items()})


def test_featnet():
    g = load_golden("net_featnet")
    sd = {k: v.requires_grad_(True) for k, v in nets.fill_params(nets.featnet_layers(), seed=3).items()}
    assert sum(v.numel() for v in sd.values()) == 136000
    out = nets.featnet_forward(sd, t(g["x"]))
    assert rel_err(out, g["out"]) < TOL
    (out * t(g["wt"])).sum().backward()
    _check_digest(g, {k: v.grad for k, v in sd.items()})


def _check_params(g, name, sd, sd0, lr, n_steps=2):
    """Post-Adam parameters against the reference run: small tensors element by element, large ones by norm at 1e-4,
    and -- the binding checks for the large ones -- the
    UPDATE p - p0 (every element moves by ~lr per step, so this is where an error in Adam or in a gradient shows):
    its norm at 1e-3 and its sum within 1e-3 of the update's norm plus an allowance for elements whose gradient is at
    rounding-noise level and may take the other sign (2*lr per step each; at most 4 + 1e-4 * numel of them)."""
    keys = [str(k) for k in g[f"p_{name}_keys"]]
    assert sorted(sd) == keys
    for i, k in enumerate(keys):
        n = float(g[f"p_{name}_norms"][i])
        assert abs(float(sd[k].double().norm()) - n) / max(n, 1e-12) < TOL, k
        upd = sd[k].double().cpu() - sd0[k].double()
        dn, ds = float(g[f"p_{name}_dnorms"][i]), float(g[f"p_{name}_dsums"][i])
        assert abs(float(upd.norm()) - dn) <= 1e-3 * dn, (k, float(upd.norm()), dn)
        flip = 2 * lr * n_steps * (4 + 1e-4 * upd.numel())
        assert abs(float(upd.sum()) - ds) <= 1e-3 * dn + flip, (k, float(upd.sum()), ds)


@pytest.mark.parametrize("with_feat", [False, True])
def test_step_unsupervise(with_feat):
    """Two full iterations (forward, losses, backward, Adam) of the unsupervise.py body."""
    g = load_golden("step_unsup_feat" if with_feat else "step_unsup")
    b, h, w = int(g["b"]), int(g["h"]), int(g["w"])
    batch = steps.synthetic_batch(b, h, w, seed=1234)
    dsd = nets.fill_params(nets.dispnet_layers(), seed=1)
    psd = nets.fill_params(nets.posenet_layers(6, 6, 2, True), seed=2)
    fsd = nets.fill_params(nets.featnet_layers(), seed=3) if with_feat else None
    st = None
    for it in range(2):
        out, grads, st = steps.step_unsupervise(dsd, psd, batch, st, feat_sd=fsd)
        assert rel_err(out["img"], g[f"img{it}"]) < TOL
        assert rel_err(out["smooth"], g[f"smooth{it}"]) < TOL
        assert rel_err(out["total"], g[f"total{it}"]) < TOL
        if with_feat:
            assert rel_err(out["feat"], g[f"feat{it}"]) < TOL
        if it == 0:
            _check_digest(g, grads["disp"], "g_disp_")
            _check_digest(g, grads["pose"], "g_pose_")
    _check_params(g, "disp", dsd, nets.fill_params(nets.dispnet_layers(), seed=1), 1e-3)
    _check_params(g, "pose", psd, nets.fill_params(nets.posenet_layers(6, 6, 2, True), seed=2), 1e-3)
    if with_feat:
        _check_params(g, "feat", fsd, nets.fill_params(nets.featnet_layers(), seed=3), 1e-3)


@pytest.mark.parametrize("name", ["step_train_sfm", "step_train_sfm_exp", "step_train_sfm_v4"])
def test_step_train_sfm(name):
    """train.py body: base case, with the explainability term (w2 = 0.2) and with nb_ref_imgs = 4 (cfg 5)."""
    g = load_golden(name)
    b, h, w, nb_ref, w2 = int(g["b"]), int(g["h"]), int(g["w"]), int(g["nb_ref"]), float(g["w2"])
    batch = steps.synthetic_batch(b, h, w, seed=1234, n_views=nb_ref)
    mk = lambda: (nets.fill_params(nets.dispnet_layers(), seed=1),
                  nets.fill_params(nets.posenet_layers(3 * (1 + nb_ref), 6 * nb_ref, nb_ref, True), seed=2))
    (dsd, psd), (dsd0, psd0) = mk(), mk()
    st = None
    for it in range(2):
        out, grads, st = steps.step_train_sfm(dsd, psd, batch, st, w2=w2, nb_ref_imgs=nb_ref)
        for k in ("photo", "smooth", "lr", "total") + (("exp",) if w2 > 0 else ()):
            assert rel_err(out[k], g[f"{k}{it}"]) < TOL, k
        if it == 0:
            _check_digest(g, grads["disp"], "g_disp_")
            _check_digest(g, grads["pose"], "g_pose_")
    _check_params(g, "disp", dsd, dsd0, 2e-4)
    _check_params(g, "pose", psd, psd0, 2e-4)


def test_compute_errors_reference_vectors():
    """Depth metrics (loss_functions_sfm.compute_errors, reference :80-116) against vectors produced by the reference
    function itself: sparse KITTI-like ground truth, Garg crop on and off, prediction clamp exercised."""
    import loss_functions_sfm as LS
    g = load_golden("metrics_compute_errors")
    for key, crop in (("crop", True), ("full", False)):
        got = LS.compute_errors(t(g["gt"]), t(g["pred"]), crop=crop)
        assert len(got) == 6
        assert np.allclose(np.array(got), g[key], rtol=1e-5, atol=1e-7), (key, got, g[key])


def test_se3_expmap():
    """Oracle SE3 exponential map (forward and the reference's hand-written backward, incl. the theta -> 0 branch)
    against vectors produced by the reference's se3_generate.py."""
    g = load_golden("se3_expmap")
    vec = t(g["vec"]).double().requires_grad_(True)
    out = geometry.se3_exp(vec)
    ref = t(g["out"])[:, 0, :3, :]
    assert rel_err(out, ref) < 1e-6
    (out * t(g["wt"])[:, 0, :3, :]).sum().backward()
    assert rel_err(vec.grad, g["g_vec"]) < 1e-5
