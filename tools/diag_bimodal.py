"""final_loss of the default bench run is bimodal (~0.0060 in 85 % of the runs, ~0.070 in the rest, also at the round-1
tree): where does the outlier trajectory leave the common one?  Runs TRIALS x 25 steps from identical initial state in
one process and prints the per-step losses of every distinct trajectory.  DVF_SERIALIZE=1: one stream."""
import argparse, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "depth-vo-feat_amd")]
import torch, bench
from dvf import lib as L
if os.environ.get("DVF_SERIALIZE") == "1":
    L.SERIALIZE = True
TRIALS = int(sys.argv[1]) if len(sys.argv) > 1 else 10
args = argparse.Namespace(batch=4, height=256, width=832, seed=0, no_graph=True, force_ddp=False, graph_ddp=False)
trajs = []
for t in range(TRIALS):
    step, fwd_bwd, opt, ddp = bench.build(args, bench.CONFIGS[2], torch.device("cuda", 0), 1, 0)
    losses = [step()[0] for _ in range(25)]
    torch.cuda.synchronize()
    trajs.append([float(x) for x in losses])
    del step, fwd_bwd, opt
    torch.cuda.empty_cache()
ref = sorted(trajs, key=lambda tr: tr[-1])[len(trajs) // 2]
for i, tr in enumerate(trajs):
    first = next((k for k, (a, b) in enumerate(zip(tr, ref)) if abs(a - b) > 1e-3 * max(abs(b), 1e-9)), None)
    print("trial %2d final %.6f first step that differs from the median trajectory by > 1e-3: %s" % (i, tr[-1], first))
    if first is not None:
        print("      ", ["%.5g" % v for v in tr[max(0, first - 2):first + 4]])
        print("   ref ", ["%.5g" % v for v in ref[max(0, first - 2):first + 4]])
