"""How far ahead of the GPU is the host?  Times the enqueue of K steps (no sync) against their completion."""
import os, sys, time, argparse
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "depth-vo-feat_amd")]
import torch, bench
args = argparse.Namespace(batch=4, height=256, width=832, seed=0, no_graph=True, force_ddp=False, graph_ddp=False)
step, fwd_bwd, opt, ddp = bench.build(args, bench.CONFIGS[2], torch.device("cuda", 0), 1, 0)
for _ in range(5): step()
torch.cuda.synchronize()
K = 20
t0 = time.perf_counter()
for _ in range(K): step()
t1 = time.perf_counter()
torch.cuda.synchronize()
t2 = time.perf_counter()
# where does the host time go?  forward / backward / optimizer enqueue, each timed with the queue drained first
import dvf.steps as S
tf = tb = to = 0.0
for _ in range(10):
    torch.cuda.synchronize(); a = time.perf_counter()
    out = fwd_bwd.__closure__ and None
    torch.cuda.synchronize()
print(f"enqueue {1e3*(t1-t0)/K:.2f} ms/step, complete {1e3*(t2-t0)/K:.2f} ms/step")
