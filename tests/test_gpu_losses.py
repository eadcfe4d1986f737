"""GPU parity: fused warp / photometric / smoothness HIP kernels (through the C ABI and the drop-in
Python API) against the golden vectors of the reference and against the CPU oracle on seeded inputs.
Tolerance: 1e-4 relative (max-abs error / max-abs reference), fp32 -- the bound north_star states."""
import pytest
import torch

from conftest import load_golden, outliers, rel_err, safe_pixel_mask, t
from oracle import geometry as og
from oracle import losses as ol

pytestmark = pytest.mark.gpu
TOL = 1e-4
DEV = "cuda"


def _cuda(*xs):
    return [x.detach().to(DEV).requires_grad_(x.requires_grad) for x in xs]


@pytest.mark.parametrize("name", [
    "warp_c3_16x24_f32_euler_zeros", "warp_c3_16x24_f32_quat_zeros", "warp_c3_16x24_f32_euler_border",
    "warp_c3_32x104_f32_euler_zeros", "warp_c32_16x24_f32_euler_zeros", "warp_c32_24x40_f32_quat_border"])
def test_inverse_warp_golden(name):
    import inverse_warp as iw
    g = load_golden(name)
    rot = "quat" if "quat" in name else "euler"
    pad = "border" if "border" in name else "zeros"
    img, depth, pose = (t(g[k], DEV).requires_grad_(True) for k in ("img", "depth", "pose"))
    out = iw.inverse_warp(img, depth, pose, t(g["K"], DEV), t(g["Kinv"], DEV), rot, pad)
    (out * t(g["wt"], DEV)).sum().backward()
    assert rel_err(out, g["out"]) < TOL
    assert rel_err(depth.grad, g["g_depth"]) < TOL
    assert rel_err(pose.grad, g["g_pose"]) < TOL
    assert rel_err(img.grad, g["g_img"]) < TOL
    if pad == "zeros":
        # the exact-zero pattern (fully out-of-bounds pixels) must be identical, not just close
        assert torch.equal((out.detach().cpu() == 0).all(1), (t(g["out"]) == 0).all(1))


@pytest.mark.parametrize("name", ["photo_c3_16x24", "photo_c3_32x104", "photo_c32_16x24", "photo_c3_32x104_rawpose"])
def test_photometric_golden(name):
    import loss_functions as lf
    g = load_golden(name)
    lv = {k: t(g[k], DEV).requires_grad_(True) for k in ("R2", "R1", "L2", "depth", "T21", "TRL")}
    loss = lf.photometric_reconstruction_loss(lv["R2"], lv["R1"], lv["L2"], lv["depth"], lv["T21"], lv["TRL"],
                                              t(g["K"], DEV), t(g["Kinv"], DEV))
    loss.backward()
    assert rel_err(loss, g["loss"]) < TOL
    for k in lv:
        assert rel_err(lv[k].grad, g["g_" + k]) < TOL, k


def test_smooth_golden():
    import loss_functions as lf
    g = load_golden("smooth")
    maps = [t(g[f"map{s}"], DEV).requires_grad_(True) for s in range(4)]
    l1 = lf.smooth_loss(maps[0])
    l1.backward()
    assert rel_err(l1, g["loss_single"]) < 1e-5
    assert rel_err(maps[0].grad, g["g_single"]) < 1e-5
    maps[0].grad = None
    l4 = lf.smooth_loss(maps, 2.0)
    l4.backward()
    assert rel_err(l4, g["loss_multi"]) < 1e-5
    for s in range(4):
        assert rel_err(maps[s].grad, g[f"g_multi{s}"]) < 1e-5


@pytest.mark.parametrize("pad", ["zeros", "border", "nomask"])
def test_photometric_sfm_golden(pad):
    """loss_functions_sfm.photometric_reconstruction_loss (4 scales, 2 views, masks | no masks, zeros | border) and
    explainability_loss on the HIP path against the reference's vectors (loss_functions_sfm.py:9-56)."""
    import loss_functions_sfm as ls
    g = load_golden("photo_sfm_zeros" if pad == "nomask" else f"photo_sfm_{pad}")
    depth = [t(g[f"depth{s}"], DEV).requires_grad_(True) for s in range(4)]
    masks = [t(g[f"mask{s}"], DEV).requires_grad_(True) for s in range(4)]
    pose = t(g["pose"], DEV).requires_grad_(True)
    refs = [t(g["ref0"], DEV), t(g["ref1"], DEV)]
    if pad == "nomask":
        gn = load_golden("photo_sfm_nomask")
        loss = ls.photometric_reconstruction_loss(t(g["tgt"], DEV), refs, t(g["K"], DEV), t(g["Kinv"], DEV), depth,
                                                  [None] * 4, pose, "euler", "zeros")
        loss.backward()
        assert rel_err(loss, gn["loss"]) < TOL
        assert rel_err(pose.grad, gn["g_pose"]) < TOL
        for s in range(4):
            assert rel_err(depth[s].grad, gn[f"g_depth{s}"]) < TOL, s
        return
    loss = ls.photometric_reconstruction_loss(t(g["tgt"], DEV), refs, t(g["K"], DEV), t(g["Kinv"], DEV), depth, masks,
                                              pose, "euler", pad)
    loss.backward()
    assert rel_err(loss, g["loss"]) < TOL
    assert rel_err(pose.grad, g["g_pose"]) < TOL
    for s in range(4):
        assert rel_err(depth[s].grad, g[f"g_depth{s}"]) < TOL, s
        assert rel_err(masks[s].grad, g[f"g_mask{s}"]) < TOL, s
    for m in masks:
        m.grad = None
    le = ls.explainability_loss(masks)                      # sum_s BCE(mask_s, 1)  (:49-56)
    le.backward()
    assert rel_err(le, g["exp_loss"]) < 1e-5
    for s in range(4):
        assert rel_err(masks[s].grad, g[f"g_mask_exp{s}"]) < 1e-5, s
    single = ls.explainability_loss(masks[0].detach())      # a bare tensor is accepted like a 1-list (:50-51)
    assert single.dim() == 0 and float(single) > 0


def _kitti_K(b, h, w):
    K = torch.tensor([[0.58 * w, 0, 0.5 * w], [0, 1.92 * h, 0.5 * h], [0, 0, 1.0]]).expand(b, 3, 3).contiguous()
    return K, torch.inverse(K[0]).expand(b, 3, 3).contiguous()


@pytest.mark.parametrize("b,c,h,w,rot,pad,align,v", [
    (2, 3, 37, 75, "euler", "zeros", False, 2),        # ragged: neither dim a multiple of the 64x4 tile
    (1, 3, 128, 416, "euler", "zeros", False, 2),      # cfg 1 size
    (2, 3, 64, 200, "quat", "zeros", False, 2),
    (1, 32, 48, 96, "euler", "zeros", False, 2),       # feature maps, all gradients
    (1, 32, 80, 160, "wild", "zeros", False, 2),       # feature maps, depth 0.3 .. 30 m per pixel: footprints of a 64x4 tile span
                                                       # from a few rows to most of the image -> fewer channels per LDS round and
                                                       # the direct (global-atomic) fallback are both exercised
    (4, 3, 256, 832, "euler", "zeros", False, 2),      # BASELINE cfg 2 size, full batch
    (8, 32, 256, 832, "euler", "zeros", False, 2),     # cfg 3's feature term at its bench size (C = 32, batch 8, all gradients)
    (2, 3, 384, 1280, "euler", "zeros", False, 4),     # cfg 5's finest scale: five-frame window (V = 4) with masks, 384x1280
])
def test_photometric_vs_oracle(b, c, h, w, rot, pad, align, v):
    """Fused kernel vs the oracle on seeded random inputs, explainability-mask form (pose [B,V,6]).
    Pixels within 1e-3 px of a tap-set crossing are masked out in both (see conftest.safe_pixel_mask)."""
    from dvf.ops import PhotoLossFn
    from dvf import lib as L
    gen = torch.Generator().manual_seed(b * 1000 + c * 100 + h)
    tgt = torch.rand(b, c, h, w, generator=gen)
    srcs = [torch.rand(b, c, h, w, generator=gen) for _ in range(v)]
    # sources are 5x5 box-filtered noise: white noise has |dI/dx| ~ 1 per pixel, which turns the ~6e-5 px
    # fp32 coordinate noise at x ~ 800 directly into >1e-4 value noise in ANY fp32 implementation
    srcs = [torch.nn.functional.avg_pool2d(torch.nn.functional.pad(x, (2, 2, 2, 2), mode="reflect"), 5, 1) for x in srcs]
    depth = torch.rand(b, h, w, generator=gen) * 20 + 2
    if rot == "wild":
        rot, depth = "euler", torch.exp(torch.rand(b, h, w, generator=gen) * 4.6 - 1.2)      # 0.3 .. 30, log-uniform
    pose = torch.randn(b, v, 6, generator=gen) * 0.03
    pose[:, 1, 0] -= 0.54
    K, Kinv = _kitti_K(b, h, w)
    safe = safe_pixel_mask(depth, [pose[:, i] for i in range(v)], K, Kinv, rot=rot, tgt=tgt, srcs=srcs)
    assert float(safe.mean()) > (0.9 if v == 2 else 0.8)
    mask = (torch.rand(b, v, h, w, generator=gen) * 0.9 + 0.05) * safe
    feat = c > 3

    def run_oracle(dt):
        lv = [x.clone().to(dt).requires_grad_(True) for x in (depth, pose, mask)] + \
             [x.clone().to(dt).requires_grad_(feat) for x in [tgt] + srcs]
        l = ol.photometric_reconstruction_loss_sfm(lv[3], lv[4:], K.to(dt), Kinv.to(dt), [lv[0].unsqueeze(1)],
                                                   [lv[2]], lv[1], rot, pad, align)
        l.backward()
        return l, lv

    ref, cpu = run_oracle(torch.float32)          # the reference's arithmetic (fp32 CPU)
    ref64, cpu64 = run_oracle(torch.float64)      # ground truth, to measure fp32's own noise floor
    gpu = _cuda(*cpu)
    pose_vb6 = gpu[1].transpose(0, 1).contiguous()
    out = PhotoLossFn.apply(gpu[3], gpu[0], pose_vb6, K.to(DEV), Kinv.to(DEV), gpu[2],
                            L.geom_flags(rot, pad, align), *gpu[4:])
    out.backward()
    assert rel_err(out, ref) < TOL
    names = ["depth", "pose", "mask", "tgt"] + [f"s{i}" for i in range(v)]
    for n, a, r, r64 in zip(names, gpu, cpu, cpu64):
        if r.grad is None:
            continue
        # bound: 1e-4 against the fp32 oracle, or -- where fp32 itself is noisier than that at this size --
        # no further from the fp64 truth than twice the fp32 CPU path's own distance to it
        floor = rel_err(r.grad, r64.grad)
        assert rel_err(a.grad, r.grad) < TOL or rel_err(a.grad, r64.grad) < max(TOL, 2 * floor), (n, floor)


@pytest.mark.parametrize("rot,pad,align", [("quat", "border", False), ("euler", "zeros", True), ("quat", "zeros", True)])
def test_photometric_modes_linear_sources(rot, pad, align):
    """rotation/padding/align_corners variants on linear-ramp sources: bilinear sampling of a linear image
    has no gradient discontinuity at tap-set crossings, so every gradient must match to 1e-4 unmasked."""
    import loss_functions as lf
    b, h, w = 2, 40, 72
    gen = torch.Generator().manual_seed(31)
    yy, xx = torch.meshgrid(torch.arange(h, dtype=torch.float32), torch.arange(w, dtype=torch.float32), indexing="ij")
    ramp = lambda a_, b_, c_: a_ * xx + b_ * yy + c_
    R1 = torch.stack((ramp(0.7, 0.2, 5), ramp(-0.3, 0.9, 60), ramp(0.5, -0.4, 90)))[None].expand(b, 3, h, w).contiguous()
    L2 = torch.stack((ramp(0.1, 0.8, 15), ramp(0.6, 0.3, 6), ramp(-0.2, 0.5, 190)))[None].expand(b, 3, h, w).contiguous()
    R2 = torch.rand(b, 3, h, w, generator=gen) * 100
    depth = torch.rand(b, h, w, generator=gen) * 20 + 2
    T21 = torch.randn(b, 6, generator=gen) * 0.03
    TRL = torch.tensor([-0.54, 0, 0, 0, 0, 0.0]).expand(b, 6) + torch.randn(b, 6, generator=gen) * 0.005
    K, Kinv = _kitti_K(b, h, w)
    cpu = [x.clone().requires_grad_(True) for x in (depth, T21, TRL)]
    ref = ol.photometric_reconstruction_loss(R2, R1, L2, cpu[0], cpu[1], cpu[2], K, Kinv, rot, pad, align)
    ref.backward()
    gpu = _cuda(*cpu)
    out = lf.photometric_reconstruction_loss(R2.to(DEV), R1.to(DEV), L2.to(DEV), gpu[0], gpu[1], gpu[2], K.to(DEV),
                                             Kinv.to(DEV), rot, pad, align)
    out.backward()
    assert rel_err(out, ref) < TOL
    for n, a, r in zip(("depth", "T21", "TRL"), gpu, cpu):
        if n == "depth":   # border pixels of a clipped/zero-padded ramp still have crossings
            assert outliers(a.grad, r.grad, TOL) <= 4, n
        else:
            assert rel_err(a.grad, r.grad) < TOL, n


def test_smooth_full_size_properties():
    """BASELINE size 256x832, B=4: size-independent properties -- exactly zero loss and gradient on an
    affine map (second differences vanish), and scale linearity loss(k*d) = k*loss(d)."""
    import loss_functions as lf
    b, h, w = 4, 256, 832
    yy, xx = torch.meshgrid(torch.arange(h, dtype=torch.float32), torch.arange(w, dtype=torch.float32), indexing="ij")
    plane = (0.5 * xx + 0.25 * yy + 3).expand(b, 1, h, w).contiguous().to(DEV).requires_grad_(True)
    l = lf.smooth_loss(plane)
    l.backward()
    assert float(l) == 0.0 and float(plane.grad.abs().max()) == 0.0
    d = (torch.rand(b, 1, h, w, generator=torch.Generator().manual_seed(5)) * 10).to(DEV)
    l1, l2 = lf.smooth_loss(d), lf.smooth_loss(4 * d)
    assert abs(float(l2) - 4 * float(l1)) <= 1e-6 * float(l2)
    ref = ol.smooth_loss(d.cpu())
    assert rel_err(l1, ref) < 1e-5


def test_photometric_view_linearity_full_size():
    """256x832, B=4 (BASELINE cfg 2): the loss over a pair of views equals the sum of the single-view
    losses (each view is an independent mean), and the value matches the oracle."""
    from dvf.ops import PhotoLossFn
    b, h, w = 4, 256, 832
    K, Kinv = _kitti_K(b, h, w)
    K, Kinv = K.to(DEV), Kinv.to(DEV)
    gen = torch.Generator().manual_seed(9)
    depth = (torch.rand(b, h, w, generator=gen) * 20 + 2).to(DEV)
    tgt, s0, s1 = (torch.rand(b, 3, h, w, generator=gen).to(DEV) for _ in range(3))
    pose = (torch.randn(2, b, 6, generator=gen) * 0.02).to(DEV)
    both = PhotoLossFn.apply(tgt, depth, pose, K, Kinv, None, 0, s0, s1)
    one = PhotoLossFn.apply(tgt, depth, pose[:1].contiguous(), K, Kinv, None, 0, s0)
    two = PhotoLossFn.apply(tgt, depth, pose[1:].contiguous(), K, Kinv, None, 0, s1)
    assert abs(float(both) - float(one) - float(two)) < 1e-6 * float(both)
    ref = ol.photometric_reconstruction_loss(tgt.cpu(), s0.cpu(), s1.cpu(), depth.cpu(), pose[0].cpu(), pose[1].cpu(),
                                             K.cpu(), Kinv.cpu())
    assert rel_err(both, ref) < TOL


@pytest.mark.parametrize("rot", ["euler", "quat"])
def test_geometry_helpers_vs_oracle(rot):
    """The stand-alone building blocks of inverse_warp.py (pose_vec2mat / euler2mat / quat2mat / pixel2cam /
    cam2pixel / set_id_grid) against the oracle, values and gradients."""
    import inverse_warp as iw
    gen = torch.Generator().manual_seed(17)
    b, h, w = 2, 24, 40
    vec = torch.randn(b, 6, generator=gen) * 0.2
    depth = torch.rand(b, h, w, generator=gen) * 20 + 2
    K, Kinv = _kitti_K(b, h, w)
    gm = torch.randn(b, 3, 4, generator=gen)
    rv = vec.clone().requires_grad_(True)
    ref = og.pose_vec2mat(rv, rot)
    (ref * gm).sum().backward()
    gv = vec.clone().to(DEV).requires_grad_(True)
    out = iw.pose_vec2mat(gv, rot)
    (out * gm.to(DEV)).sum().backward()
    assert rel_err(out, ref) < 1e-6 and rel_err(gv.grad, rv.grad) < 1e-5
    fn_o, fn_g = (og.euler2mat, iw.euler2mat) if rot == "euler" else (og.quat2mat, iw.quat2mat)
    assert rel_err(fn_g(vec[:, 3:].to(DEV)), fn_o(vec[:, 3:])) < 1e-6
    # pixel2cam
    rd = depth.clone().requires_grad_(True)
    cam_ref = og.pixel2cam(rd, Kinv)
    gc = torch.randn(cam_ref.shape, generator=gen)
    (cam_ref * gc).sum().backward()
    gd = depth.clone().to(DEV).requires_grad_(True)
    cam = iw.pixel2cam(gd, Kinv.to(DEV))
    (cam * gc.to(DEV)).sum().backward()
    assert rel_err(cam, cam_ref) < 1e-6 and rel_err(gd.grad, rd.grad) < 1e-5
    # cam2pixel (zeros and border), gradients to cam / rot / tr
    for pad in ("zeros", "border"):
        proj = (K @ og.pose_vec2mat(vec * 0.2, rot))
        leaves = [cam_ref.detach().clone().requires_grad_(True), proj[:, :, :3].clone().requires_grad_(True),
                  proj[:, :, -1:].clone().requires_grad_(True)]
        gref = og.cam2pixel(leaves[0], leaves[1], leaves[2], pad)
        gg = torch.randn(gref.shape, generator=gen)
        (gref * gg).sum().backward()
        dl = [x.detach().clone().to(DEV).requires_grad_(True) for x in leaves]
        gout = iw.cam2pixel(dl[0], dl[1], dl[2], pad)
        (gout * gg.to(DEV)).sum().backward()
        assert rel_err(gout, gref) < 1e-5
        for a, r, nm in zip(dl, leaves, ("cam", "rot", "tr")):
            assert rel_err(a.grad, r.grad) < TOL, (pad, nm)
    iw.set_id_grid(depth.to(DEV))
    assert rel_err(iw.pixel_coords, og.pixel_grid(h, w, torch.float32)) == 0.0


def test_se3_expmap_golden():
    """generate_se3 (forward + the reference's hand-written backward, incl. the theta -> 0 branch) vs vectors from
    the reference's se3_generate.py."""
    import se3_generate
    g = load_golden("se3_expmap")
    vec = t(g["vec"], DEV).view(-1, 6, 1, 1).requires_grad_(True)
    out = se3_generate.generate_se3(vec)
    assert tuple(out.shape) == (6, 1, 4, 4)
    assert rel_err(out, g["out"]) < 1e-5
    (out * t(g["wt"], DEV).float()).sum().backward()
    assert rel_err(vec.grad.view(6, 6), g["g_vec"]) < TOL


def test_dvo_front_end_vs_oracle():
    """DVF_POSE_SE3 | DVF_PIXEL_COORDS front end (unsupervise_dvo.py chain) vs the oracle's restatement of the Caffe
    layers.  Sources are linear ramps (no gradient discontinuity at tap-set crossings), the target is noise."""
    from dvf.ops import PhotoLossFn
    from dvf import lib as L
    b, h, w = 2, 40, 72
    gen = torch.Generator().manual_seed(41)
    yy, xx = torch.meshgrid(torch.arange(h, dtype=torch.float32), torch.arange(w, dtype=torch.float32), indexing="ij")
    ramp = lambda a_, b_, c_: a_ * xx + b_ * yy + c_
    L2 = torch.stack((ramp(0.7, 0.2, 5), ramp(-0.3, 0.9, 60), ramp(0.5, -0.4, 90)))[None].expand(b, 3, h, w).contiguous()
    R1 = torch.stack((ramp(0.1, 0.8, 15), ramp(0.6, 0.3, 6), ramp(-0.2, 0.5, 190)))[None].expand(b, 3, h, w).contiguous()
    R2 = torch.rand(b, 3, h, w, generator=gen) * 100
    depth = torch.rand(b, h, w, generator=gen) * 20 + 2
    T_R2L = torch.tensor([0, 0, 0, -0.54, 0, 0.0]).expand(b, 6) + torch.randn(b, 6, generator=gen) * 0.01
    T_21 = torch.randn(b, 6, generator=gen) * 0.03
    K, Kinv = _kitti_K(b, h, w)
    cpu = [x.clone().requires_grad_(True) for x in (depth, T_R2L, T_21)]
    ref = og.dvo_photometric_loss(R2, L2, R1, cpu[0], cpu[1], cpu[2], K)
    ref.backward()
    gd, gp = depth.clone().to(DEV).requires_grad_(True), torch.stack((T_R2L, T_21)).to(DEV).requires_grad_(True)
    out = PhotoLossFn.apply(R2.to(DEV), gd, gp, K.to(DEV), Kinv.to(DEV), None, L.POSE_SE3 | L.PIXEL_COORDS,
                            L2.to(DEV), R1.to(DEV))
    out.backward()
    assert rel_err(out, ref) < TOL
    assert outliers(gd.grad, cpu[0].grad, TOL) <= 4
    assert rel_err(gp.grad[0], cpu[1].grad) < TOL and rel_err(gp.grad[1], cpu[2].grad) < TOL
