"""Which parameters' gradients differ bitwise between two runs of the same step from identical state?  (cfg-2 body)"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, "depth-vo-feat_amd")); sys.path.insert(0, ROOT)
import torch
import DispNetS, PoseExpNet
from dvf.engine import FlatAdam
from dvf import conv as C
C.set_deterministic(os.environ.get('DET', '1') != '0')
from dvf.steps import unsupervise_losses
from dvf.synthetic import synthetic_batch
b, h, w = (int(x) for x in (sys.argv[1:4] if len(sys.argv) > 3 else (4, 256, 832)))
torch.manual_seed(0)
disp, pose = DispNetS.DispNetS(), PoseExpNet.PoseExpNet(output_exp=True)
disp.init_weights(); pose.init_weights()
disp.cuda().train(); pose.cuda().train()
opt = FlatAdam(list(pose.parameters()) + list(disp.parameters()), lr=1e-3, weight_decay=1e-8)
batch = synthetic_batch(b, h, w, seed=1234, device="cuda")
arenas = []
for rep in range(3):
    loss, terms = unsupervise_losses(disp, pose, batch)
    opt.zero_grad()
    loss.backward()
    opt.join_wgrad()
    from dvf import lib as L
    L.join_aux_streams()
    torch.cuda.synchronize()
    arenas.append(opt.flat_g.clone())
names = {id(p): n for m, tag in ((disp, "disp"), (pose, "pose")) for n, p in ((f"{tag}.{k}", v) for k, v in m.named_parameters())}
bad = []
for p, o in zip(opt.params, opt.offsets):
    n = p.numel()
    d = [bool((arenas[0][o:o + n] != arenas[i][o:o + n]).any()) for i in (1, 2)]
    if any(d):
        bad.append(names[id(p)])
print("parameters whose gradient differs between runs: %d of %d" % (len(bad), len(opt.params)))
print(" ".join(bad))
