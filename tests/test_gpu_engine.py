"""GPU tests of the training-step engine (dvf/engine.py) beyond the golden steps:
 * a module applied TWICE per step (shared weights: two gradient contributions per parameter) must not let its bucket
   leave -- all-reduce, Adam, repack -- after the first contribution: three iterations against the CPU oracle, with the
   early per-bucket update and the exchange path on;
 * packed weight copies used by EAGER code between HIP-graph replays (validate() with a batch size the capture never saw)
   must follow the weights the replays update."""
import pytest
import torch

from oracle import nets as onets
from oracle import steps as osteps

pytestmark = pytest.mark.gpu
DEV = "cuda"


def _load(module, sd):
    res = module.load_state_dict({k: v.clone() for k, v in sd.items()}, strict=True)
    assert not res.missing_keys and not res.unexpected_keys
    return module.to(DEV)


def _oracle_twice(sd, x1, x2, w1, w2, steps, lr):
    """loss = sum_i <w1_i, disp_i(x1)> + <w2_i, disp_i(x2)> with ONE DispNetS, Adam -- plain torch CPU autograd."""
    params = osteps._leaf(sd)
    state = osteps.adam_init(params)
    losses = []
    for _ in range(steps):
        for p in params.values():
            p.grad = None
        o1, o2 = onets.dispnet_forward(params, x1), onets.dispnet_forward(params, x2)
        loss = sum((o * w).sum() for o, w in zip(o1, w1)) + sum((o * w).sum() for o, w in zip(o2, w2))
        loss.backward()
        losses.append(float(loss))
        with torch.no_grad():
            osteps.adam_step(params, {k: p.grad for k, p in params.items()}, state, lr)
    return losses, {k: v.detach() for k, v in params.items()}


@pytest.mark.parametrize("early", [True, False])
def test_module_used_twice_per_step_tracks_oracle(early):
    import DispNetS
    from dvf.engine import FlatAdam
    torch.manual_seed(3)
    h, w, steps, lr = 64, 128, 3, 1e-3
    sd = onets.fill_params(onets.dispnet_layers(), seed=5)
    x1, x2 = torch.rand(1, 3, h, w), torch.rand(1, 3, h, w)
    shapes = [(1, 1, h >> s, w >> s) for s in range(4)]
    w1 = [torch.randn(sh) * 1e-3 for sh in shapes]
    w2 = [torch.randn(sh) * 1e-3 for sh in shapes]
    ref_losses, ref_params = _oracle_twice(sd, x1, x2, w1, w2, steps, lr)

    net = _load(DispNetS.DispNetS(), sd)
    net.train()
    # small buckets: several of them, so that a premature launch would hit layers whose second contribution is pending
    opt = FlatAdam(list(net.parameters()), lr=lr, bucket_mb=4.0, overlap=True, always_reduce=False, early_update=early)
    gx1, gx2 = x1.to(DEV), x2.to(DEV)
    gw1, gw2 = [v.to(DEV) for v in w1], [v.to(DEV) for v in w2]
    losses, launched_in_backward = [], []
    for _ in range(steps):
        o1, o2 = net(gx1), net(gx2)
        loss = sum((o * v).sum() for o, v in zip(o1, gw1)) + sum((o * v).sum() for o, v in zip(o2, gw2))
        opt.zero_grad()
        loss.backward()
        launched_in_backward.append(sum(1 for b in opt.buckets if b["launched"]))
        opt.step()
        losses.append(float(loss))
    torch.cuda.synchronize()
    for p in opt.params:
        assert p._dvf_expect == 2, "every DispNetS parameter receives two contributions per step"
    if early:
        assert launched_in_backward[0] == 0 and launched_in_backward[-1] == len(opt.buckets), launched_in_backward
    for a, b in zip(losses, ref_losses):
        assert abs(a - b) <= 1e-4 * max(abs(b), 1e-6), (losses, ref_losses)
    # Adam turns noise-level gradients into O(lr) steps: compare the displacement of each tensor
    for k, v in net.state_dict().items():
        d_ref = (ref_params[k] - sd[k]).double()
        d_new = (v.detach().cpu() - sd[k]).double()
        assert float(d_ref.norm()) > 0
        assert float((d_new - d_ref).norm()) <= 0.05 * float(d_ref.norm()) + 1e-9, k


def test_extra_contribution_after_a_bucket_left_is_refused():
    """More grad_ready() calls than the previous step made, arriving after the bucket has left: an error, not silence."""
    from dvf.engine import FlatAdam
    ps = [torch.nn.Parameter(torch.zeros(1000, device=DEV)) for _ in range(3)]
    opt = FlatAdam(ps, lr=1e-3, overlap=True, early_update=True)
    for step in range(2):
        opt.zero_grad()
        for p in opt.params:
            opt.grad_ready(p)
        opt.step()
    opt.zero_grad()
    for p in opt.params:
        opt.grad_ready(p)                    # the (single) bucket leaves here
    assert opt.buckets[0]["launched"]
    with pytest.raises(RuntimeError, match="after its bucket had left"):
        opt.grad_ready(opt.params[0])
    opt.step()
    opt.relearn()                            # one step without early launches, then the new pattern is the learnt one
    for step in range(3):
        opt.zero_grad()
        for p in opt.params:
            opt.grad_ready(p)
            if step > 0:
                assert not opt.buckets[0]["launched"]
            opt.grad_ready(p)
        opt.step()
    assert all(p._dvf_expect == 2 for p in opt.params)


def test_eager_convolutions_between_graph_replays_see_fresh_weights():
    """validate() runs eagerly between replays; a geometry first seen there (a last partial batch) gets a packed
    copy whose stamp the replays never invalidate.  The eager output after N replays must equal the eager output of a model
    trained WITHOUT the graph for N steps."""
    import PoseExpNet
    from dvf.engine import FlatAdam, GraphedStep
    torch.manual_seed(4)
    h, w = 64, 128
    sd = onets.fill_params(onets.posenet_layers(6, 6, 2, True), seed=7)
    xt = torch.rand(2, 6, h, w, device=DEV)
    xv = torch.rand(1, 6, h, w, device=DEV)             # "validation": batch 1, never captured
    results = {}
    for mode in ("graph", "eager"):
        net = _load(PoseExpNet.PoseExpNet(output_exp=True), sd)
        net.train()
        opt = FlatAdam(list(net.parameters()), lr=1e-2)

        def step():
            masks, pose = net(xt)
            loss = pose.square().sum() + sum(m.mean() for m in masks)
            opt.zero_grad()
            loss.backward()
            opt.step()
            return (loss,)

        def validate():
            net.eval()
            with torch.no_grad():
                _, pose = net(xv)
            net.train()
            torch.cuda.synchronize()
            return pose.clone()

        vals = []
        if mode == "graph":
            runner = GraphedStep(step, [], warmup=1)     # 1 eager step + capture (the capture pass does not execute)
            vals.append(validate())
            for _ in range(2):
                runner()
                vals.append(validate())
        else:
            step()
            vals.append(validate())
            for _ in range(2):
                step()
                vals.append(validate())
        results[mode] = vals
    for i, (a, b) in enumerate(zip(results["graph"], results["eager"])):
        assert float((a - b).abs().max()) <= 1e-5 * max(float(b.abs().max()), 1e-6), (i, a, b)
    # and the weights did move between validations (otherwise the test shows nothing)
    assert float((results["eager"][0] - results["eager"][2]).abs().max()) > 1e-4 * float(results["eager"][0].abs().max())


def test_deterministic_mode_gradient_arena_is_bit_identical():
    """dvf.conv.set_deterministic(True): three runs of the cfg-2 step from identical state leave bit-identical gradient arenas
    (the reference's CPU path is run-to-run deterministic; the default build sums weight gradients with float atomics).  Run on
    the three-stream schedule the benchmark uses, at a size where the Stream-K split spreads every layer over many blocks."""
    import DispNetS
    import PoseExpNet
    from dvf import conv as C
    from dvf import lib as L
    from dvf.engine import FlatAdam
    from dvf.steps import unsupervise_losses
    from dvf.synthetic import synthetic_batch
    torch.manual_seed(0)
    disp, pose = DispNetS.DispNetS(), PoseExpNet.PoseExpNet(output_exp=True)
    disp.init_weights(); pose.init_weights()
    disp.to(DEV).train(); pose.to(DEV).train()
    opt = FlatAdam(list(pose.parameters()) + list(disp.parameters()), lr=1e-3, weight_decay=1e-8)
    batch = synthetic_batch(2, 128, 416, seed=1234, device=DEV)
    C.set_deterministic(True)
    try:
        arenas = []
        for _ in range(3):
            loss, _terms = unsupervise_losses(disp, pose, batch)
            opt.zero_grad()
            loss.backward()
            opt.join_wgrad()
            L.join_aux_streams()
            torch.cuda.synchronize()
            arenas.append(opt.flat_g.clone())
    finally:
        C.set_deterministic(False)
    assert float(arenas[0].abs().max()) > 0
    assert torch.equal(arenas[0], arenas[1]) and torch.equal(arenas[0], arenas[2])
