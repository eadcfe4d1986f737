"""GPU parity: fused warp / photometric / smoothness HIP kernels (through the C ABI and the drop-in
Python API) against the golden vectors of the reference and against the CPU oracle on seeded inputs.
Tolerance: 1e-4 relative (max-abs error / max-abs reference), fp32 -- the bound north_star states."""
import pytest
import torch

from conftest import load_golden, rel_err, t
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


@pytest.mark.parametrize("name", ["photo_c3_16x24", "photo_c3_32x104", "photo_c32_16x24"])
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


@pytest.mark.parametrize("b,c,h,w,rot,pad,align", [
    (2, 3, 37, 75, "euler", "zeros", False),        # ragged: neither dim a multiple of the 64x4 tile
    (1, 3, 128, 416, "euler", "zeros", False),      # cfg 1 size
    (2, 3, 64, 200, "quat", "border", False),
    (2, 3, 40, 72, "euler", "zeros", True),         # align_corners=True opt-in
    (1, 32, 48, 96, "euler", "zeros", False),       # feature maps, all gradients
])
def test_photometric_vs_oracle(b, c, h, w, rot, pad, align):
    import loss_functions as lf
    gen = torch.Generator().manual_seed(b * 1000 + c * 100 + h)
    R2, R1, L2 = (torch.rand(b, c, h, w, generator=gen) for _ in range(3))
    depth = torch.rand(b, h, w, generator=gen) * 20 + 2
    T21 = torch.randn(b, 6, generator=gen) * 0.03
    TRL = torch.tensor([-0.54, 0, 0, 0, 0, 0.0]).expand(b, 6) + torch.randn(b, 6, generator=gen) * 0.005
    K = torch.tensor([[0.58 * w, 0, 0.5 * w], [0, 1.92 * h, 0.5 * h], [0, 0, 1.0]]).expand(b, 3, 3).contiguous()
    Kinv = torch.inverse(K[0]).expand(b, 3, 3).contiguous()
    feat = c > 3
    cpu = [x.clone().requires_grad_(True) for x in (depth, T21, TRL)] + \
          [x.clone().requires_grad_(feat) for x in (R2, R1, L2)]
    ref = ol.photometric_reconstruction_loss(cpu[3], cpu[4], cpu[5], cpu[0], cpu[1], cpu[2], K, Kinv, rot, pad, align)
    ref.backward()
    gpu = _cuda(*cpu)
    out = lf.photometric_reconstruction_loss(gpu[3], gpu[4], gpu[5], gpu[0], gpu[1], gpu[2], K.to(DEV), Kinv.to(DEV),
                                             rot, pad, align)
    out.backward()
    assert rel_err(out, ref) < TOL
    names = ["depth", "T21", "TRL", "R2", "R1", "L2"]
    for n, a, r in zip(names, gpu, cpu):
        if r.grad is not None:
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


def test_photometric_full_size_properties():
    """256x832, B=4: identity pose + source == target gives a loss that only comes from the half-pixel
    shift of align_corners=False sampling of a constant image (== 0 for constant images), and the loss of
    a pair of views is the sum of the single-view losses (linearity over views)."""
    from dvf.ops import PhotoLossFn
    b, h, w = 4, 256, 832
    K = torch.tensor([[0.58 * w, 0, 0.5 * w], [0, 1.92 * h, 0.5 * h], [0, 0, 1.0]]).expand(b, 3, 3).contiguous()
    Kinv = torch.inverse(K[0]).expand(b, 3, 3).contiguous().to(DEV)
    K = K.to(DEV)
    gen = torch.Generator().manual_seed(9)
    depth = (torch.rand(b, h, w, generator=gen) * 20 + 2).to(DEV)
    const = torch.full((b, 3, h, w), 0.7, device=DEV)
    zero_pose = torch.zeros(1, b, 6, device=DEV)
    l = PhotoLossFn.apply(const, depth, zero_pose, K, Kinv, None, 0, const)
    assert float(l) < 1e-6
    tgt, s0, s1 = (torch.rand(b, 3, h, w, generator=gen).to(DEV) for _ in range(3))
    pose = (torch.randn(2, b, 6, generator=gen) * 0.02).to(DEV)
    both = PhotoLossFn.apply(tgt, depth, pose, K, Kinv, None, 0, s0, s1)
    one = PhotoLossFn.apply(tgt, depth, pose[:1].contiguous(), K, Kinv, None, 0, s0)
    two = PhotoLossFn.apply(tgt, depth, pose[1:].contiguous(), K, Kinv, None, 0, s1)
    assert abs(float(both) - float(one) - float(two)) < 1e-6 * float(both)
    ref = ol.photometric_reconstruction_loss(tgt.cpu(), s0.cpu(), s1.cpu(), depth.cpu(), pose[0].cpu(), pose[1].cpu(),
                                             K.cpu(), Kinv.cpu())
    assert rel_err(both, ref) < TOL
