"""GPU parity of the fp32-MFMA convolution kernels (forward, dgrad, wgrad, fused bias/activation, virtual
concat, crop_like, parity-class transposed convolution, split-K) and of the small memory-bound ops, against
torch CPU fp32 (the arithmetic the reference's nn.Conv2d / nn.ConvTranspose2d / F.interpolate run on).
Bound: 1e-4 relative (max-abs error / max-abs reference)."""
import pytest
import torch
import torch.nn.functional as F

from conftest import rel_err

pytestmark = pytest.mark.gpu
TOL = 1e-4
DEV = "cuda"


def _ref_act(x, act, alpha, beta):
    if act == 1:
        return F.relu(x)
    if act == 2:
        return alpha * torch.sigmoid(x) + beta
    return x


CASES = [
    # name, segs, cout, k, stride, pad, opad, transposed, act, (N,H,W), out_hw
    ("c3x3_s1_relu", [16], 32, 3, 1, 1, 0, False, 1, (2, 20, 36), None),
    ("c3x3_s2_relu", [16], 64, 3, 2, 1, 0, False, 1, (2, 21, 37), None),
    ("c7x7_s2_cin3", [3], 32, 7, 2, 3, 0, False, 1, (2, 32, 48), None),
    ("c7x7_s1", [32], 32, 7, 1, 3, 0, False, 1, (1, 24, 40), None),
    ("c5x5_s2", [32], 64, 5, 2, 2, 0, False, 1, (2, 24, 40), None),
    ("c5x5_s1", [64], 64, 5, 1, 2, 0, False, 1, (1, 12, 20), None),
    ("c3x3_concat3", [64, 64, 1], 64, 3, 1, 1, 0, False, 1, (2, 16, 26), None),
    ("c3x3_concat2_odd", [16, 1], 16, 3, 1, 1, 0, False, 1, (1, 32, 52), None),
    ("head_sigmoid_affine", [16], 1, 3, 1, 1, 0, False, 2, (2, 32, 52), None),
    ("head2_sigmoid_ragged", [19], 2, 3, 1, 1, 0, False, 2, (2, 21, 70), None),
    ("head4_relu", [8], 4, 3, 1, 1, 0, False, 1, (1, 9, 130), None),
    ("mask_head_sigmoid", [32], 2, 3, 1, 1, 0, False, 2, (2, 16, 26), None),
    ("pose_pred_1x1", [256], 12, 1, 1, 0, 0, False, 0, (2, 2, 7), None),
    ("deep_splitk", [512], 512, 3, 1, 1, 0, False, 1, (2, 4, 13), None),
    ("deep_s2_tiny", [512], 512, 3, 2, 1, 0, False, 1, (2, 4, 13), None),
    ("t3x3_s2_op1", [64], 32, 3, 2, 1, 1, True, 1, (2, 16, 26), None),
    ("t3x3_s2_op1_crop", [512], 512, 3, 2, 1, 1, True, 1, (2, 2, 7), (4, 13)),
    ("t4x4_s2", [64], 32, 4, 2, 1, 0, True, 1, (2, 8, 13), None),
    ("t4x4_s2_crop", [32], 16, 4, 2, 1, 0, True, 1, (1, 16, 26), (31, 51)),
    ("c3x3_wide", [32], 32, 3, 1, 1, 0, False, 1, (1, 40, 208), None),
    # thin full-resolution layers (iconv1): direct kernel, forward over the virtual concat + dgrad of the 16-channel segment
    ("thin16_concat2_ragged", [16, 1], 16, 3, 1, 1, 0, False, 1, (2, 70, 75), None),
    ("thin16_concat3", [8, 13, 3], 16, 3, 1, 1, 0, False, 1, (1, 64, 66), None),
    ("thin16_single_noact", [16], 16, 3, 1, 1, 0, False, 0, (1, 65, 64), None),
    # weight gradients with at most 16 rows: 16x16x4 MFMA tiles in wgrad_pipe (NTW 1..4, strides 1 / 2, both DMA widths)
    ("wg16_c3x3", [5], 16, 3, 1, 1, 0, False, 1, (2, 24, 40), None),
    ("wg16_c3x3_wide", [28], 12, 3, 1, 1, 0, False, 1, (1, 20, 37), None),
    ("wg16_c7x7_s2", [6], 16, 7, 2, 3, 0, False, 1, (2, 32, 48), None),
    ("wg16_c5x5_s2_odd", [9], 8, 5, 2, 2, 0, False, 0, (1, 23, 41), None),
    # thin stride-2 transposed convolutions (upconv1 / upconv2 of both networks): direct forward kernel
    ("dconvt3_16", [32], 16, 3, 2, 1, 1, True, 1, (2, 64, 66), None),
    ("dconvt3_16_crop", [64], 16, 3, 2, 1, 1, True, 1, (1, 64, 72), (127, 143)),
    ("dconvt4_16", [32], 16, 4, 2, 1, 0, True, 1, (1, 64, 64), None),
    ("dconvt4_16_crop_odd", [24], 16, 4, 2, 1, 0, True, 0, (1, 65, 70), (129, 139)),
]


@pytest.mark.parametrize("case", CASES, ids=[c[0] for c in CASES])
def test_conv_fwd_bwd(case):
    from dvf.conv import ConvFn
    name, segs, cout, k, stride, pad, opad, transposed, act, (n, h, w), out_hw = case
    gen = torch.Generator().manual_seed(hash(name) % 10000)
    cin = sum(segs)
    xs = [torch.randn(n, c, h, w, generator=gen) for c in segs]
    wshape = (cin, cout, k, k) if transposed else (cout, cin, k, k)
    wt = torch.randn(wshape, generator=gen) / (cin * k * k) ** 0.5
    b = torch.randn(cout, generator=gen) * 0.1
    alpha, beta = (10.0, 0.01) if "affine" in name else (1.0, 0.0)
    # ---- reference: torch CPU
    rx = [x.clone().requires_grad_(True) for x in xs]
    rw, rb = wt.clone().requires_grad_(True), b.clone().requires_grad_(True)
    xin = torch.cat(rx, 1)
    if transposed:
        pre = F.conv_transpose2d(xin, rw, rb, stride=stride, padding=pad, output_padding=opad)
    else:
        pre = F.conv2d(xin, rw, rb, stride=stride, padding=pad)
    if out_hw is not None:
        pre = pre[:, :, :out_hw[0], :out_hw[1]]
    ref = _ref_act(pre, act, alpha, beta)
    gout = torch.randn(ref.shape, generator=gen)
    if act == 1:
        # ReLU kink: an output whose pre-activation is within rounding of 0 may land on either side in two
        # correct fp32 implementations (different summation order); such elements get no upstream gradient
        gout = gout * (pre.detach().abs() > 1e-4)
    (ref * gout).sum().backward()
    # ---- HIP
    gx = [x.clone().to(DEV).requires_grad_(True) for x in xs]
    gw, gb = wt.clone().to(DEV).requires_grad_(True), b.clone().to(DEV).requires_grad_(True)
    out = ConvFn.apply(gw, gb, (k, stride, pad, opad, transposed, act, alpha, beta, out_hw), *gx)
    assert out.shape == ref.shape
    (out * gout.to(DEV)).sum().backward()
    assert rel_err(out, ref) < TOL, "forward"
    assert rel_err(gw.grad, rw.grad) < TOL, "wgrad"
    assert rel_err(gb.grad, rb.grad) < TOL, "bias grad"
    for i, (a, r) in enumerate(zip(gx, rx)):
        assert rel_err(a.grad, r.grad) < TOL, f"dgrad seg {i}"


def test_conv_input_without_grad_skips_dgrad():
    from dvf.conv import ConvFn
    x = torch.randn(1, 3, 16, 16, device=DEV)
    w = torch.randn(8, 3, 3, 3, device=DEV, requires_grad=True)
    b = torch.zeros(8, device=DEV, requires_grad=True)
    out = ConvFn.apply(w, b, (3, 1, 1, 0, False, 1, 1.0, 0.0, None), x)
    out.sum().backward()
    assert x.grad is None and w.grad is not None


@pytest.mark.parametrize("n,c,h,w,out_hw", [(2, 1, 16, 26, (32, 52)), (2, 1, 4, 13, (7, 25)), (1, 3, 5, 7, (10, 14))])
def test_upsample2x(n, c, h, w, out_hw):
    from dvf.conv import Upsample2xFn
    gen = torch.Generator().manual_seed(3)
    x = torch.randn(n, c, h, w, generator=gen)
    rx = x.clone().requires_grad_(True)
    ref = F.interpolate(rx, scale_factor=2, mode="bilinear", align_corners=False)[:, :, :out_hw[0], :out_hw[1]]
    g = torch.randn(ref.shape, generator=gen)
    (ref * g).sum().backward()
    gx = x.clone().to(DEV).requires_grad_(True)
    out = Upsample2xFn.apply(gx, out_hw)
    (out * g.to(DEV)).sum().backward()
    assert rel_err(out, ref) < 1e-6 and rel_err(gx.grad, rx.grad) < 1e-6


def test_small_ops():
    from dvf import conv as C
    gen = torch.Generator().manual_seed(4)
    x = torch.rand(2, 1, 16, 26, generator=gen) * 10 + 0.01
    rx = x.clone().requires_grad_(True)
    ref = 1 / (rx + 1e-4)
    g = torch.randn(ref.shape, generator=gen)
    (ref * g).sum().backward()
    gx = x.clone().to(DEV).requires_grad_(True)
    out = C.reciprocal(gx, 1e-4)
    (out * g.to(DEV)).sum().backward()
    assert rel_err(out, ref) < 1e-6 and rel_err(gx.grad, rx.grad) < 1e-6
    # spatial mean
    p = torch.randn(2, 12, 2, 7, generator=gen)
    rp = p.clone().requires_grad_(True)
    rm = 0.01 * rp.mean(3).mean(2)
    gm = torch.randn(rm.shape, generator=gen)
    (rm * gm).sum().backward()
    gp = p.clone().to(DEV).requires_grad_(True)
    om = C.SpatialMeanFn.apply(gp, 0.01)
    (om * gm.to(DEV)).sum().backward()
    assert rel_err(om, rm) < 1e-6 and rel_err(gp.grad, rp.grad) < 1e-6
    # area / bilinear-half pyramids (no gradient)
    img = torch.rand(2, 3, 32, 64, generator=gen) * 255
    for s in (2, 4, 8):
        assert rel_err(C.area_downsample(img.to(DEV), (32 // s, 64 // s)), F.interpolate(img, (32 // s, 64 // s), mode="area")) < 1e-6
    assert rel_err(C.area_downsample(img.to(DEV), (10, 21)), F.interpolate(img, (10, 21), mode="area")) < 1e-6
    assert rel_err(C.bilinear_half(img.to(DEV)), F.interpolate(img, scale_factor=0.5, mode="bilinear")) < 1e-6


FUSE_CASES = [
    # name, cin, cmid, (N,H,W): a ReLU layer A (fuse_bwd) feeding three consumers
    ("fuse_shallow", 16, 64, (2, 20, 36)),       # pipelined dgrad epilogue + stride-2 classes + head kernel post-pass
    ("fuse_deep_splitk", 256, 512, (2, 4, 13)),  # split-K finish with the mask, transposed consumer
    ("fuse_thin", 8, 16, (1, 33, 50)),           # unpacked kernels -> mask pass after them, odd sizes
]


@pytest.mark.parametrize("case", FUSE_CASES, ids=[c[0] for c in FUSE_CASES])
def test_fused_relu_backward(case):
    """ReLU backward + bias gradient of a producing layer done by its consumers' dgrad kernels (dvf_conv2d_dgrad_masked):
    y = relu(convA(x)) feeds a 3x3 conv, a stride-2 conv / transposed conv and a 1-channel head; every gradient is
    compared with torch CPU, and with the unfused path (FUSE_RELU_BWD = False) of this library."""
    from dvf import conv as C
    from dvf import lib as L
    name, cin, cmid, (n, h, w) = case
    gen = torch.Generator().manual_seed(321 + cin)
    x = torch.randn(n, cin, h, w, generator=gen)

    def build(fuse):
        g2 = torch.Generator().manual_seed(99)
        A = C.FusedConv2d(cin, cmid, 3, 1, 1, L.ACT_RELU, fuse_bwd=fuse)
        B = C.FusedConv2d(cmid, 64, 3, 1, 1, L.ACT_RELU)
        Cc = C.FusedConvTranspose2d(cmid, 32, 3, 2, 1, L.ACT_RELU, output_padding=1) if cmid >= 256 else \
            C.FusedConv2d(cmid, 32, 3, 2, 1, L.ACT_RELU)
        D = C.FusedConv2d(cmid, 1, 3, 1, 1, L.ACT_SIGMOID_AFFINE, alpha=10.0, beta=0.01)
        mods = [A, B, Cc, D]
        for m in mods:
            with torch.no_grad():
                m.weight.copy_(torch.randn(m.weight.shape, generator=g2) / (m.weight[0].numel()) ** 0.5)
                m.bias.copy_(torch.randn(m.bias.shape, generator=g2) * 0.1)
        return mods

    def run(mods, dev, functional):
        A, B, Cc, D = mods
        xi = x.clone().to(dev).requires_grad_(True)
        if functional:        # torch reference on CPU
            y = F.relu(F.conv2d(xi, A.weight, A.bias, padding=1))
            o1 = F.relu(F.conv2d(y, B.weight, B.bias, padding=1))
            if Cc.transposed:
                o2 = F.relu(F.conv_transpose2d(y, Cc.weight, Cc.bias, stride=2, padding=1, output_padding=1))
            else:
                o2 = F.relu(F.conv2d(y, Cc.weight, Cc.bias, stride=2, padding=1))
            o3 = 10.0 * torch.sigmoid(F.conv2d(y, D.weight, D.bias, padding=1)) + 0.01
        else:
            y = A(xi)
            o1, o2, o3 = B(y), Cc(y), D(y)
        gg = torch.Generator().manual_seed(5)
        loss = sum((o * torch.randn(o.shape, generator=gg).to(dev)).sum() for o in (o1, o2, o3))
        loss.backward()
        grads = [xi.grad] + [p.grad for m in mods for p in (m.weight, m.bias)]
        return [g.detach().cpu() for g in grads]

    ref = run(build(False), "cpu", True)
    fused_mods = [m.to(DEV) for m in build(True)]
    got = run(fused_mods, DEV, False)
    assert getattr(fused_mods[0], "fuse_bwd") is True
    C.FUSE_RELU_BWD = False
    try:
        plain = run([m.to(DEV) for m in build(True)], DEV, False)
    finally:
        C.FUSE_RELU_BWD = True
    names = ["x"] + [f"{m}.{p}" for m in "ABCD" for p in ("weight", "bias")]
    for nm, r, g, pl in zip(names, ref, got, plain):
        assert rel_err(g, r) < TOL, (name, nm, "fused vs torch", rel_err(g, r))
        assert rel_err(pl, r) < TOL, (name, nm, "unfused vs torch", rel_err(pl, r))
