"""Randomised envelope tests of the convolution paths (weight gradients: wgrad_pipe.hip + its fallbacks; forward and (masked) dgrad): 48 seeded geometries -- kernel
sizes 1..7, strides 1/2, Conv2d and ConvTranspose2d, 1..3 input segments, odd sizes down to 2x3 pixels, channel counts on
both sides of the 16 / 32 / 64-row tile boundaries, with and without the fused bias column -- against torch CPU.
Bound: 1e-4 relative (max-abs error / max-abs reference), fp32."""
import random

import pytest
import torch
import torch.nn.functional as F

from conftest import rel_err

pytestmark = pytest.mark.gpu
TOL = 1e-4
DEV = "cuda"


def _cases():
    rng = random.Random(20261004)
    out = []
    for i in range(48):
        k = rng.choice([1, 3, 3, 3, 4, 5, 7])
        stride = rng.choice([1, 1, 2])
        transposed = rng.random() < 0.3 and k >= 3
        pad = (k - 1) // 2 if not transposed else 1
        opad = 1 if (transposed and stride == 2 and k == 3) else 0
        if transposed and stride == 1:
            pad, opad = (k - 1) // 2, 0
        nseg = 1 if transposed else rng.choice([1, 1, 2, 3])
        segs = [rng.choice([1, 3, 5, 8, 16, 17, 31, 33, 64]) for _ in range(nseg)]
        cout = rng.choice([2, 6, 12, 16, 17, 32, 40, 64, 70])
        n = rng.choice([1, 2, 3])
        h, w = rng.choice([2, 4, 7, 13, 20, 33]), rng.choice([3, 7, 13, 24, 26, 36, 52, 65])
        if not transposed and (h + 2 * pad < k or w + 2 * pad < k):
            h, w = max(h, k), max(w, k)
        out.append((f"fz{i}_k{k}s{stride}{'T' if transposed else 'C'}_{'+'.join(map(str, segs))}->{cout}_{n}x{h}x{w}",
                    segs, cout, k, stride, pad, opad, transposed, (n, h, w), rng.random() < 0.5))
    return out


CASES = _cases()


@pytest.mark.parametrize("case", CASES, ids=[c[0] for c in CASES])
def test_wgrad_envelope(case):
    from dvf.conv import ConvFn, ReluTag
    from dvf import lib as L
    name, segs, cout, k, stride, pad, opad, transposed, (n, h, w), fused = case
    gen = torch.Generator().manual_seed(hash(name) % 100000)
    cin = sum(segs)
    xs = [torch.randn(n, c, h, w, generator=gen) for c in segs]
    wshape = (cin, cout, k, k) if transposed else (cout, cin, k, k)
    wt = torch.randn(wshape, generator=gen) / (cin * k * k) ** 0.5
    b = torch.randn(cout, generator=gen) * 0.1
    rw, rb = wt.clone().requires_grad_(True), b.clone().requires_grad_(True)
    xin = torch.cat(xs, 1)
    if transposed:
        pre = F.conv_transpose2d(xin, rw, rb, stride=stride, padding=pad, output_padding=opad)
    else:
        pre = F.conv2d(xin, rw, rb, stride=stride, padding=pad)
    gout = torch.randn(pre.shape, generator=gen)
    (pre * gout).sum().backward()
    gx = [x.clone().to(DEV) for x in xs]
    gw, gb = wt.clone().to(DEV).requires_grad_(True), b.clone().to(DEV).requires_grad_(True)
    # fused: the layer is a ReLU layer whose consumers delivered dL/dpre (here: gout masked by relu') and whose bias gradient
    # rides on the weight-gradient launch; plain: no activation, bias gradient by the reduction pass
    act = L.ACT_RELU if fused else L.ACT_NONE
    cfg = (k, stride, pad, opad, transposed, act, 1.0, 0.0, None) + ((ReluTag(),) if fused else ())
    out = ConvFn.apply(gw, gb, cfg, *gx)
    assert tuple(out.shape) == tuple(pre.shape), (name, out.shape, pre.shape)
    out.backward(gout.to(DEV))
    assert rel_err(gw.grad, rw.grad) < TOL, (name, "wgrad", rel_err(gw.grad, rw.grad))
    assert rel_err(gb.grad, rb.grad) < TOL, (name, "bias", rel_err(gb.grad, rb.grad))


@pytest.mark.parametrize("case", CASES, ids=[c[0] for c in CASES])
def test_fwd_dgrad_envelope(case):
    """Forward (bias + ReLU / none fused in the specialised epilogues) and the input gradient -- plain, or masked by the
    segments' own sign when they are tagged as outputs of ReLU layers -- over the same geometries."""
    from dvf.conv import ConvFn, ReluTag
    from dvf import lib as L
    name, segs, cout, k, stride, pad, opad, transposed, (n, h, w), fused = case
    gen = torch.Generator().manual_seed(hash(name) % 100000 + 7)
    cin = sum(segs)
    xs = [torch.randn(n, c, h, w, generator=gen) for c in segs]
    if fused:
        xs = [F.relu(x) for x in xs]
    wshape = (cin, cout, k, k) if transposed else (cout, cin, k, k)
    wt = torch.randn(wshape, generator=gen) / (cin * k * k) ** 0.5
    b = torch.randn(cout, generator=gen) * 0.1
    rx = [x.clone().requires_grad_(True) for x in xs]
    xin = torch.cat(rx, 1)
    if transposed:
        pre = F.conv_transpose2d(xin, wt, b, stride=stride, padding=pad, output_padding=opad)
    else:
        pre = F.conv2d(xin, wt, b, stride=stride, padding=pad)
    ref = F.relu(pre)
    gout = torch.randn(ref.shape, generator=gen) * (pre.detach().abs() > 1e-4)
    (ref * gout).sum().backward()
    gx = [x.clone().to(DEV).requires_grad_(True) for x in xs]
    if fused:
        for x in gx:
            x._dvf_relu_tag = ReluTag()
    gw, gb = wt.clone().to(DEV).requires_grad_(True), b.clone().to(DEV).requires_grad_(True)
    out = ConvFn.apply(gw, gb, (k, stride, pad, opad, transposed, L.ACT_RELU, 1.0, 0.0, None), *gx)
    assert rel_err(out, ref) < TOL, (name, "fwd", rel_err(out, ref))
    out.backward(gout.to(DEV))
    for i, (a, r, x0) in enumerate(zip(gx, rx, xs)):
        want = r.grad * (x0 > 0) if fused else r.grad
        assert rel_err(a.grad, want) < TOL, (name, f"dgrad{i}", rel_err(a.grad, want))
