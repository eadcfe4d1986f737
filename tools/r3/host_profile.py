"""Where does the HOST spend a training step?  cProfile over eager steps of the cfg-2 body at a tiny image size (the GPU work is
negligible there, the host work per step is the same as at full size).  Backward runs on the calling thread so that the
profiler sees ConvFn.backward."""
import cProfile, io, os, pstats, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, "depth-vo-feat_amd")); sys.path.insert(0, ROOT)
import torch
import DispNetS, PoseExpNet
from dvf.engine import FlatAdam
from dvf.steps import unsupervise_losses
from dvf.synthetic import synthetic_batch
torch.autograd.set_multithreading_enabled(False)
disp, pose = DispNetS.DispNetS(), PoseExpNet.PoseExpNet(output_exp=True)
disp.init_weights(); pose.init_weights(); disp.cuda().train(); pose.cuda().train()
opt = FlatAdam(list(pose.parameters()) + list(disp.parameters()), lr=1e-3, weight_decay=1e-8)
batch = synthetic_batch(1, 64, 128, seed=1234, device="cuda")
def step():
    loss, terms = unsupervise_losses(disp, pose, batch)
    opt.zero_grad(); loss.backward(); opt.step()
for _ in range(5): step()
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(20): step()
torch.cuda.synchronize()
print("ms/step (single-threaded autograd): %.3f" % ((time.perf_counter() - t0) / 20 * 1e3))
pr = cProfile.Profile(); pr.enable()
for _ in range(20): step()
pr.disable(); torch.cuda.synchronize()
s = io.StringIO(); pstats.Stats(pr, stream=s).sort_stats("tottime").print_stats(45); print(s.getvalue()[:9000])
s = io.StringIO(); pstats.Stats(pr, stream=s).sort_stats("cumulative").print_stats(30); print(s.getvalue()[:6000])
