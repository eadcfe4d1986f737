"""The data-parallel exchange path on real hardware with a world-size-1 RCCL group (row a14 of SURVEY.md section 8):
buckets all-reduced on the communication stream from inside backward, behind the weight-gradient side streams and the
pose network's auxiliary stream, then Adam with the 1/world factor.  A self all-reduce is the identity, so the golden
training step must come out exactly as without the exchange -- any missing stream dependency of the overlapped path
shows up as a mismatch against the reference run.  (N > 1 runs only on the driver's 8-GPU node; the bookkeeping for
world 2 is covered on CPU by tests/test_ddp_gloo.py.)"""
import os
import socket

import pytest
import torch
import torch.distributed as dist

from conftest import load_golden, rel_err
from oracle import nets as onets
from test_gpu_nets import _batch, _check_digest, _check_params, _init, _load

pytestmark = pytest.mark.gpu
TOL = 1e-4
DEV = "cuda"


@pytest.fixture(scope="module")
def nccl_world1():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    os.environ["MASTER_ADDR"], os.environ["MASTER_PORT"] = "127.0.0.1", str(port)
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda", 0))
    yield
    dist.destroy_process_group()


@pytest.mark.parametrize("overlap", [True, False])
def test_golden_step_through_the_exchange_path(nccl_world1, overlap):
    import DispNetS
    import PoseExpNet
    from dvf.engine import FlatAdam
    from dvf.steps import unsupervise_losses
    g = load_golden("step_unsup")
    b, h, w = int(g["b"]), int(g["h"]), int(g["w"])
    batch = _batch(b, h, w)
    disp = _load(DispNetS.DispNetS(), _init("disp"))
    pose = _load(PoseExpNet.PoseExpNet(output_exp=True), _init("pose", 6, 6, 2, True))
    disp.train(); pose.train()
    opt = FlatAdam(list(pose.parameters()) + list(disp.parameters()), lr=1e-3, weight_decay=1e-8, world_size=1,
                   always_reduce=True, overlap=overlap, bucket_mb=25.0)
    assert len(opt.buckets) >= 5 and opt.exchange
    launched_in_backward = []
    for it in range(2):
        loss, terms = unsupervise_losses(disp, pose, batch)
        opt.zero_grad()
        loss.backward()
        launched_in_backward.append(sum(1 for bk in opt.buckets if bk["launched"]))
        if it == 0:
            opt.join_wgrad()
            _check_digest(g, {k: p.grad for k, p in disp.named_parameters() if p.grad is not None}, "g_disp_")
            _check_digest(g, {k: p.grad for k, p in pose.named_parameters() if p.grad is not None}, "g_pose_")
        opt.step()
        for k in ("img", "smooth", "total"):
            assert rel_err(terms[k], g[f"{k}{it}"]) < TOL, k
    torch.cuda.synchronize()
    assert opt.n_reduced == 2 * len(opt.buckets)
    # first step: the set of parameters that receive a gradient is unknown -> exchanged in step(); from the second step
    # on every bucket goes out from inside backward when the exchange is overlapped
    assert launched_in_backward == ([0, len(opt.buckets)] if overlap else [0, 0])
    _check_params(g, "disp", disp, _init("disp"), 1e-3)
    _check_params(g, "pose", pose, _init("pose", 6, 6, 2, True), 1e-3)
