"""Parity AT THE SHAPES THE BENCHMARK RUNS (cfg 2: 256x832, batch 4).

The convolution planner is shape dependent (tile arrangement, NT, stage count / LDS budget, split-K factor, producer
waves, wgrad channel chunks and vector paths all derive from N*H*W), so the toy-size cases of test_gpu_conv.py do not
prove the kernel variants the headline number runs on.  Here:

  1. one full-size training step (DispNetS + PoseExpNet, photometric V=2 + 10*smooth) is compared with the CPU oracle
     -- per-term losses and every parameter's gradient -- while every convolution geometry of the step and every plan
     record the library reports (dvf_conv2d_last_plans) is collected;
  2. every distinct convolution geometry of that step is run alone (forward, dgrad of the segments that need it, wgrad,
     bias grad) against torch CPU conv2d / conv_transpose2d at 1e-4;
  3. the set of plans hit in 2 must contain the set the step used in 1.
Bound: 1e-4 relative (max-abs error / max-abs reference) per layer; for the step, losses at 1e-4 and gradients by
norm at 1e-4, and by norm of the difference at max(2e-4, 8 x the fp32 oracle's own distance from an fp64 run of the
oracle): composite gradients of the deep layers are ill-conditioned in fp32 for ANY implementation (the CPU oracle is
2.5e-4 away from fp64 at conv7), so the bound is stated relative to that measured floor; a wrong kernel is O(1) off."""
import pytest
import torch
import torch.nn.functional as F

from conftest import rel_err
from oracle import nets as onets
from oracle import steps as osteps

pytestmark = pytest.mark.gpu
TOL = 1e-4
DEV = "cuda"
B, H, W = 4, 256, 832
STATE = {}


def _ref_act(x, act, alpha, beta):
    if act == 1:
        return F.relu(x)
    if act == 2:
        return alpha * torch.sigmoid(x) + beta
    return x


def _step_nets():
    import DispNetS
    import PoseExpNet
    dsd = onets.fill_params(onets.dispnet_layers(), seed=1)
    psd = onets.fill_params(onets.posenet_layers(6, 6, 2, True), seed=2)
    disp, pose = DispNetS.DispNetS(), PoseExpNet.PoseExpNet(output_exp=True)
    disp.load_state_dict({k: v.clone() for k, v in dsd.items()})
    pose.load_state_dict({k: v.clone() for k, v in psd.items()})
    return dsd, psd, disp.to(DEV).train(), pose.to(DEV).train()


def test_fullsize_step_vs_oracle():
    """cfg-2 step at the benchmark size through FlatAdam's arena (weight gradients accumulate in place on the side
    streams, exactly as bench.py runs it) vs oracle.steps.step_unsupervise(do_update=False)."""
    from dvf import conv as C
    from dvf import lib as L
    from dvf.engine import FlatAdam
    from dvf.steps import unsupervise_losses
    from dvf.synthetic import synthetic_batch
    dsd, psd, disp, pose = _step_nets()
    opt = FlatAdam(list(pose.parameters()) + list(disp.parameters()), lr=1e-3, weight_decay=1e-8)
    batch = synthetic_batch(B, H, W, seed=1234, device=DEV)
    C.GEOM_LOG, L.PLAN_LOG = [], set()
    try:
        loss, terms = unsupervise_losses(disp, pose, batch)
        opt.zero_grad()
        loss.backward()
        opt.join_wgrad()
        L.join_aux_streams()
        torch.cuda.synchronize()
    finally:
        STATE["geoms"], STATE["step_plans"] = C.GEOM_LOG, L.PLAN_LOG
        C.GEOM_LOG, L.PLAN_LOG = None, None
    ref, grads, _ = osteps.step_unsupervise(dsd, psd, osteps.synthetic_batch(B, H, W, seed=1234), do_update=False)
    # fp64 run of the same oracle: ground truth for the noise floor of fp32 itself (ReLU gates and bilinear tap sets of
    # near-degenerate pixels flip under a different summation order; the deep 2x7 layers collect that noise)
    d64 = lambda sd: {k: v.double() for k, v in sd.items()}
    _, grads64, _ = osteps.step_unsupervise(d64(dsd), d64(psd), osteps.synthetic_batch(B, H, W, seed=1234, dtype=torch.float64),
                                            do_update=False)
    for k in ("img", "smooth", "total"):
        assert rel_err(terms[k], ref[k]) < TOL, (k, float(terms[k]), float(ref[k]))
    worst = (0.0, 0.0, "")
    for name, mod in (("disp", disp), ("pose", pose)):
        for k, p in mod.named_parameters():
            if k not in grads[name]:
                assert p.grad is None or float(p.grad.abs().max()) == 0.0, k
                continue
            r = grads[name][k].double()
            g = p.grad.detach().double().cpu()
            rn = max(float(r.norm()), 1e-30)
            e_norm, e_diff = abs(float(g.norm()) - rn) / rn, float((g - r).norm()) / rn
            worst = max(worst, (e_diff, e_norm, f"{name}.{k}"))
            r64 = grads64[name][k]
            floor, e64 = float((r - r64).norm()) / rn, float((g - r64).norm()) / rn
            assert e_norm < TOL or abs(float(g.norm()) - float(r64.norm())) / rn < max(TOL, 2 * abs(float(r.norm()) - float(r64.norm())) / rn), (name, k, e_norm)
            # direction: 2e-4, or 8x the distance of the fp32 ORACLE ITSELF from the fp64 truth for this parameter.
            # Round 3 measured what that factor is (profiles/r03_gradient_noise_seeds.txt, r03_noise_knobs.txt,
            # r03_blocked_accumulation.txt): with the SAME kernels the ratio |g - g64| / |g32 - g64| of the deep layers is
            # 0.3 ... 8 depending on the data / weight seed alone, and on this seed it moves 0.9 <-> 9 with the summation
            # blocking of the forward / dgrad kernels although every layer's own rms error stays at or below torch's CPU
            # kernels' (1.4-2.0e-7).  It counts which of two fp32 runs drew the worse set of ReLU-gate / bilinear-cell flips;
            # it cannot be tightened to 2x by better arithmetic.  A wrong kernel is O(1) off.
            assert e_diff < max(2e-4, 8 * floor) and e64 < max(2e-4, 8 * floor), (name, k, e_diff, e64, floor)
    print("worst gradient: |g-ref|/|ref| = %.2e, norm error %.2e at %s" % worst)


def _distinct(geoms):
    seen, out = set(), []
    for g in geoms:
        if g not in seen:
            seen.add(g)
            out.append(g)
    return out


def _check_geometry(gi, geom, failures):
    """Forward / dgrad / wgrad / bias-grad of one convolution geometry against torch CPU at 1e-4: plain, and in the fused
    form the steps run (masked dgrad, weight gradient + bias column)."""
    from dvf.conv import ConvFn, ReluTag
    segs, cout, cfg, (n, h, w), need_in = geom
    k, stride, pad, opad, transposed, act, alpha, beta, out_hw = cfg
    gen = torch.Generator().manual_seed(1000 + gi)
    cin = sum(segs)
    xs = [torch.randn(n, c, h, w, generator=gen) for c in segs]
    wshape = (cin, cout, k, k) if transposed else (cout, cin, k, k)
    wt = torch.randn(wshape, generator=gen) / (cin * k * k) ** 0.5
    b = torch.randn(cout, generator=gen) * 0.1
    rx = [x.clone().requires_grad_(ng) for x, ng in zip(xs, need_in)]
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
        gout = gout * (pre.detach().abs() > 1e-4)       # ReLU kink: see test_gpu_conv.py
    (ref * gout).sum().backward()
    gx = [x.clone().to(DEV).requires_grad_(ng) for x, ng in zip(xs, need_in)]
    gw, gb = wt.clone().to(DEV).requires_grad_(True), b.clone().to(DEV).requires_grad_(True)
    out = ConvFn.apply(gw, gb, cfg, *gx)
    (out * gout.to(DEV)).sum().backward()
    tag = f"{'T' if transposed else 'C'}{k}x{k}s{stride} {list(segs)}->{cout} @{h}x{w} N{n}"
    errs = {"fwd": rel_err(out, ref), "wgrad": rel_err(gw.grad, rw.grad), "bias": rel_err(gb.grad, rb.grad)}
    for i, (a, r) in enumerate(zip(gx, rx)):
        if r.grad is not None:
            errs[f"dgrad{i}"] = rel_err(a.grad, r.grad)
    bad = {kk: v for kk, v in errs.items() if not v < TOL}
    if bad or tuple(out.shape) != tuple(ref.shape):
        failures.append((tag, bad))
    del out, gx, gw, gb, rx, rw, rb, ref, pre, gout
    # ---- the same geometry as the step runs it (dvf/conv.py::ReluTag): inputs are outputs of ReLU layers, so the
    # dgrad leaves multiplied by (x > 0); a ReLU layer itself receives dL/dpre (its consumers masked it) and its
    # bias gradient is the extra column of the weight-gradient kernel (dvf_conv2d_wgrad_bias)
    xs2 = [F.relu(x) for x in xs]
    rx = [x.clone().requires_grad_(ng) for x, ng in zip(xs2, need_in)]
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
        gout = gout * (pre.detach().abs() > 1e-4)
    (ref * gout).sum().backward()
    gx = [x.clone().to(DEV).requires_grad_(ng) for x, ng in zip(xs2, need_in)]
    for x in gx:
        x._dvf_relu_tag = ReluTag()
    gw, gb = wt.clone().to(DEV).requires_grad_(True), b.clone().to(DEV).requires_grad_(True)
    out = ConvFn.apply(gw, gb, tuple(cfg) + ((ReluTag(),) if act == 1 else ()), *gx)
    go = gout * (ref.detach() > 0) if act == 1 else gout        # what masking consumers deliver
    out.backward(go.to(DEV))
    errs = {"fused wgrad": rel_err(gw.grad, rw.grad), "fused bias": rel_err(gb.grad, rb.grad)}
    for i, (a, r, x0) in enumerate(zip(gx, rx, xs2)):
        if r.grad is not None:
            errs[f"masked dgrad{i}"] = rel_err(a.grad, r.grad * (x0 > 0))
    bad = {kk: v for kk, v in errs.items() if not v < TOL}
    if bad:
        failures.append((tag, bad))
    del out, gx, gw, gb, rx, rw, rb, ref, pre, gout, go


def _check_geometries(geoms, seed_base):
    """Run the geometries not parity-checked yet (any configuration); their plan records join STATE['layer_plans']."""
    from dvf import lib as L
    done = STATE.setdefault("checked_geoms", set())
    plans = STATE.setdefault("layer_plans", set())
    L.PLAN_LOG = set()
    failures = []
    try:
        for gi, geom in enumerate(geoms):
            if geom in done:
                continue
            _check_geometry(seed_base + gi, geom, failures)
            done.add(geom)
            torch.cuda.empty_cache()
    finally:
        plans |= L.PLAN_LOG
        L.PLAN_LOG = None
    assert not failures, failures


def test_every_bench_layer_geometry():
    """Forward / dgrad / wgrad / bias-grad of every distinct convolution geometry of the cfg-2 step at batch 4."""
    if "geoms" not in STATE:
        test_fullsize_step_vs_oracle()
    geoms = _distinct(STATE["geoms"])
    assert len(geoms) >= 40, len(geoms)            # DispNetS: 14 + 7 + 7 + 4, PoseExpNet: 7 + 1 + 5 + 4 (some coincide)
    _check_geometries(geoms, 0)


def test_bench_plans_are_the_tested_plans():
    """Every (op, kernel, tiling) record of the full-size step also ran in a per-layer parity case."""
    if "layer_plans" not in STATE:
        test_every_bench_layer_geometry()
    step, layer = STATE["step_plans"], STATE["layer_plans"]
    assert step, "the library reported no plans"
    missing = sorted(step - layer)
    assert not missing, missing
    kernels = {p[1] for p in step}
    assert {1, 8} <= kernels, kernels                      # conv_pipe and wgrad_pipe at least
    print("%d distinct plans in the step, all parity-tested; kernels used: %s" % (len(step), sorted(kernels)))


# ---------------------------------------------------------------------------------------------------- cfg 3 / 4 / 5
# The other benchmark configurations (bench.CONFIGS): FeatExtractor on 3 x 8 full-resolution images
# (reference: unsupervise.py:104-111), the 4-scale train.py body with the feature term, and 384x1280 with a five-frame
# window (PoseExpNet_sfm.py:22-36: a 15-channel first convolution; loss_functions_sfm.py:9-46 with V = 4).  A full-size
# ORACLE step of these is minutes of CPU time per configuration; the body arithmetic is pinned at reduced size
# (tests/test_gpu_nets.py, golden steps).  What is shape dependent -- the convolution plans and the fused loss kernels'
# size-dependent paths -- is pinned here: one full-size step per configuration collects every convolution geometry and
# every plan record; each geometry not seen before is parity-checked against torch CPU at 1e-4 (plain and fused forms);
# the step's plans must be a subset of the parity-checked plans; the step's loss must be finite.
def _bench_module():
    import importlib.util
    import os
    from conftest import ROOT
    spec = importlib.util.spec_from_file_location("dvf_bench_for_tests", os.path.join(ROOT, "bench.py"))
    m = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(m)
    return m


@pytest.mark.parametrize("config", [3, 4, 5])
def test_other_bench_configs_geometries_and_plans(config):
    import argparse
    from dvf import conv as C
    from dvf import lib as L
    bench = _bench_module()
    cfg = bench.CONFIGS[config]
    args = argparse.Namespace(batch=cfg["batch"], height=cfg["height"], width=cfg["width"], seed=0, force_ddp=False,
                              no_graph=True)
    step, fwd_bwd, opt, _ = bench.build(args, cfg, torch.device(DEV), 1, 0)
    fwd_bwd()                                         # (first pass: packs, workspaces)
    C.GEOM_LOG, L.PLAN_LOG = [], set()
    try:
        total = fwd_bwd()[0]
        L.join_aux_streams()
        torch.cuda.synchronize()
    finally:
        geoms, step_plans = _distinct(C.GEOM_LOG), L.PLAN_LOG
        C.GEOM_LOG, L.PLAN_LOG = None, None
    assert torch.isfinite(total).all(), float(total)
    assert float(opt.flat_g.abs().max()) > 0 and torch.isfinite(opt.flat_g).all()
    del step, fwd_bwd, opt
    torch.cuda.empty_cache()
    assert geoms and step_plans
    _check_geometries(geoms, 10000 * config)
    missing = sorted(step_plans - STATE["layer_plans"])
    assert not missing, (config, missing)
    print("cfg %d: %d geometries, %d plans, all parity-tested" % (config, len(geoms), len(step_plans)))
