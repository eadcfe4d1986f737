"""Diagnostic (GPU box): eager-vs-eager and eager-vs-graph parameter divergence after 4 Adam steps."""
import sys, os, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "depth-vo-feat_amd"))
from oracle import nets as onets
import DispNetS, PoseExpNet
from dvf.engine import FlatAdam, GraphedStep
from dvf.steps import unsupervise_losses
from dvf.synthetic import synthetic_batch
b, h, w = 1, 64, 128
batch = synthetic_batch(b, h, w, seed=1234, device="cuda")
def run(mode):
    disp = DispNetS.DispNetS(); disp.load_state_dict(onets.fill_params(onets.dispnet_layers(), seed=1)); disp.cuda()
    pose = PoseExpNet.PoseExpNet(); pose.load_state_dict(onets.fill_params(onets.posenet_layers(6, 6, 2, True), seed=2)); pose.cuda()
    opt = FlatAdam(list(pose.parameters()) + list(disp.parameters()), lr=1e-3, weight_decay=1e-8)
    def step():
        loss, terms = unsupervise_losses(disp, pose, batch)
        opt.zero_grad(); loss.backward(); opt.step()
        return (terms["total"],)
    if mode == "graph":
        r = GraphedStep(step, [], warmup=2); [r() for _ in range(2)]
    else:
        [step() for _ in range(4)]
    torch.cuda.synchronize()
    names = {}
    for net, nm in ((disp, "disp"), (pose, "pose")):
        for k, p in net.named_parameters():
            names[nm + "." + k] = (p.detach().clone(), p._dvf_grad.clone(), p._dvf_touched)
    return names
a, b2, c = run("eager"), run("eager"), run("graph")
for tag, x, y in (("eager-eager", a, b2), ("eager-graph", a, c)):
    worst = sorted(((float((x[k][0] - y[k][0]).abs().max()), k) for k in x), reverse=True)[:6]
    print(tag, [(f"{v:.2e}", k, f"gradmax={float(x[k][1].abs().max()):.2e}", x[k][2]) for v, k in worst])
