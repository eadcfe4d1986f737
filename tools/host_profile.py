import os, sys, time, argparse, cProfile, pstats
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "depth-vo-feat_amd")]
import torch, bench
args = argparse.Namespace(batch=4, height=256, width=832, seed=0, no_graph=True, force_ddp=False, graph_ddp=False)
step, fwd_bwd, opt, ddp = bench.build(args, bench.CONFIGS[2], torch.device("cuda", 0), 1, 0)
for _ in range(5): step()
torch.cuda.synchronize()
pr = cProfile.Profile(); pr.enable()
for _ in range(10): step()
pr.disable(); torch.cuda.synchronize()
st = pstats.Stats(pr); st.sort_stats("tottime").print_stats(28)
