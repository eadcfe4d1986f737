"""Which torch ops of a training step launch stock kernels (copies, fills, adds)?  torch.profiler over one step."""
import os, sys, argparse
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "depth-vo-feat_amd")]
import torch, bench
from torch.profiler import profile, ProfilerActivity
args = argparse.Namespace(batch=4, height=256, width=832, seed=0, no_graph=True, force_ddp=False, graph_ddp=False)
step, fwd_bwd, opt, ddp = bench.build(args, bench.CONFIGS[2], torch.device("cuda", 0), 1, 0)
for _ in range(3): step()
torch.cuda.synchronize()
with profile(activities=[ProfilerActivity.CPU], record_shapes=True, with_stack=True) as prof:
    step()
    torch.cuda.synchronize()
rows = [e for e in prof.key_averages(group_by_input_shape=True) if e.key.startswith("aten::") and e.key not in ("aten::empty", "aten::empty_like", "aten::view", "aten::as_strided", "aten::empty_strided", "aten::detach", "aten::alias", "aten::slice", "aten::select", "aten::unsqueeze", "aten::squeeze", "aten::reshape", "aten::_unsafe_view", "aten::expand", "aten::is_nonzero", "aten::item", "aten::_local_scalar_dense")]
rows.sort(key=lambda e: -e.count)
for e in rows[:40]:
    print("%-28s x%-3d shapes %s" % (e.key, e.count, str(e.input_shapes)[:110]))
