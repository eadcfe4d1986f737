"""Data-parallel exchange of dvf.engine.FlatAdam with world_size 2 over gloo on CPU tensors: arena layout,
bucket slicing, readiness counting (buckets launch from inside backward from the second step on), sum
all-reduce, and the 1/world averaging convention.  The Adam kernel itself needs the GPU and is not called."""
import os
import socket
import sys

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from conftest import PKG, ROOT


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _worker(rank, world, port, q):
    for p in (ROOT, PKG):
        sys.path.insert(0, p)
    os.environ["MASTER_ADDR"], os.environ["MASTER_PORT"] = "127.0.0.1", str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from dvf.engine import FlatAdam
        torch.manual_seed(0)                                   # identical parameters on every rank
        params = [torch.nn.Parameter(torch.randn(n)) for n in (1000, 37, 5000, 64, 3000, 11)]
        before = [p.detach().clone() for p in params]
        opt = FlatAdam(params, lr=1e-3, world_size=world, bucket_mb=0.01)      # ~2.6k floats per bucket
        assert len(opt.buckets) >= 3
        # arena views keep the values and are 64-float aligned
        for p, b in zip(params, before):
            assert torch.equal(p.detach(), b)
        assert all(o % 64 == 0 for o in opt.offsets)
        launched_in_backward = []
        for step in range(3):
            opt.zero_grad()
            # "backward": gradients become ready in arena order (reverse registration); the last parameter of the
            # list (index 5) is never touched -> it must be skipped by the exchange bookkeeping
            for p in opt.params:
                if p is params[5]:
                    continue
                p._dvf_grad.add_(torch.full_like(p, float(rank + 1) * (step + 1)))
                opt.grad_ready(p)
            launched_in_backward.append(sum(1 for b in opt.buckets if b["launched"]))
            opt.synchronize_grads()
            # sum over ranks of (rank+1)*(step+1) = 3*(step+1); Adam divides by world afterwards
            for p in params[:5]:
                assert torch.allclose(p.grad, torch.full_like(p, 3.0 * (step + 1))), (rank, step)
            assert params[5].grad is None and not params[5]._dvf_touched
        # first step: the touched set is unknown -> everything is exchanged at synchronize_grads();
        # later steps: every touched bucket is launched from inside "backward"
        assert launched_in_backward[0] == 0
        touched_buckets = sum(1 for b in opt.buckets if any(p._dvf_touched for p in b["params"]))
        assert launched_in_backward[1] == launched_in_backward[2] == touched_buckets
        rngs = opt._touched_ranges()
        assert sum(e - o for o, e in rngs) == sum((p.numel() + 63) // 64 * 64 for p in params[:5])
        q.put((rank, "ok"))
    except Exception as e:                                     # noqa: BLE001
        import traceback
        q.put((rank, f"{type(e).__name__}: {e} | " + traceback.format_exc().splitlines()[-2].strip()))
    finally:
        dist.destroy_process_group()


def test_flat_arena_exchange_world2():
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    results = [q.get(timeout=120) for _ in procs]
    for p in procs:
        p.join(timeout=60)
    assert sorted(results) == [(0, "ok"), (1, "ok")], results


class _FakeStream:
    """Stands in for torch.cuda.Stream on a CPU box: records what it was told to wait for."""

    def __init__(self, name, device="cpu"):
        self.name, self.device, self.waited = name, device, []

    def wait_stream(self, other):
        self.waited.append(other.name)


def _worker_overlap(rank, world, port, q):
    for p in (ROOT, PKG):
        sys.path.insert(0, p)
    os.environ["MASTER_ADDR"], os.environ["MASTER_PORT"] = "127.0.0.1", str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        import contextlib
        from dvf import lib as L
        from dvf.engine import FlatAdam
        torch.manual_seed(0)
        params = [torch.nn.Parameter(torch.randn(n)) for n in (3000, 3000, 3000)]
        opt = FlatAdam(params, lr=1e-3, world_size=world, bucket_mb=0.01)
        assert len(opt.buckets) == 3
        # mock the GPU streams of the overlapped exchange: communication stream, the step's main stream, the stream a
        # backward node runs on, one weight-gradient side stream, one auxiliary (pose network) stream
        comm, main, cur, side, aux = (_FakeStream(n) for n in ("comm", "main", "current", "side", "aux"))
        opt._comm_stream, opt._main_stream, opt._sides = comm, main, {1: side}
        L.AUX_STREAMS["cpu"] = aux
        order = []
        real_all_reduce = dist.all_reduce

        def spy_all_reduce(t, **kw):
            order.append(("all_reduce", tuple(comm.waited)))
            return real_all_reduce(t, **kw)

        torch.cuda.current_stream = lambda *a, **k: cur
        torch.cuda.stream = lambda s: contextlib.nullcontext()
        dist.all_reduce = spy_all_reduce
        for step in range(2):
            comm.waited.clear()
            order.clear()
            opt.zero_grad()
            opt._main_stream = main                             # (zero_grad records it only for GPU arenas)
            for p in opt.params:
                p._dvf_grad.add_(float(rank + 1))
                opt.grad_ready(p)
            launched = len(order)
            opt.synchronize_grads()
            assert cur.waited[-1] == "comm"                     # the compute stream joins the exchange before Adam
            # step 0: the touched set is unknown during backward -> nothing launches early; step 1: every bucket does
            assert launched == (0 if step == 0 else 3), (step, launched)
            assert len(order) == 3
            for _, waited in order:                             # every exchange waited for ALL compute streams first
                assert {"current", "main", "side", "aux"} <= set(waited), waited
            for p in params:
                assert torch.allclose(p.grad, torch.full_like(p, 3.0))
        q.put((rank, "ok"))
    except Exception as e:                                     # noqa: BLE001
        import traceback
        q.put((rank, f"{type(e).__name__}: {e} | " + " / ".join(traceback.format_exc().splitlines()[-4:])))
    finally:
        dist.destroy_process_group()


def test_overlapped_exchange_waits_for_every_compute_stream_world2():
    """The bucket launched from inside backward must wait for the main stream, the stream of the backward node, the
    weight-gradient side streams and the auxiliary stream before its all-reduce (streams mocked on CPU; the same code
    path runs on the GPU under tests/test_gpu_ddp.py)."""
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker_overlap, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    results = [q.get(timeout=120) for _ in procs]
    for p in procs:
        p.join(timeout=60)
    assert sorted(results) == [(0, "ok"), (1, "ok")], results


def test_weak_scaling_gradient_identity():
    """every loss term is a mean over B*C*H*W including masked pixels, so with equal local batches
    the rank-average of local-mean gradients equals the global-batch gradient.  Checked with the CPU oracle."""
    from oracle import losses as ol
    g = torch.Generator().manual_seed(0)
    b, h, w = 4, 16, 24
    tgt, s0, s1 = (torch.rand(b, 3, h, w, generator=g) for _ in range(3))
    depth = (torch.rand(b, h, w, generator=g) * 10 + 2).requires_grad_(True)
    p0 = (torch.randn(b, 6, generator=g) * 0.02)
    p1 = torch.tensor([-0.54, 0, 0, 0, 0, 0.0]).expand(b, 6).contiguous()
    K = torch.tensor([[0.58 * w, 0, 0.5 * w], [0, 1.92 * h, 0.5 * h], [0, 0, 1.0]]).expand(b, 3, 3).contiguous()
    Kinv = torch.inverse(K[0]).expand(b, 3, 3).contiguous()
    full = ol.photometric_reconstruction_loss(tgt, s0, s1, depth, p0, p1, K, Kinv) + ol.smooth_loss(depth.unsqueeze(1))
    full.backward()
    g_full = depth.grad.clone()
    depth.grad = None
    halves = []
    for sl in (slice(0, 2), slice(2, 4)):
        d = depth[sl]
        l = ol.photometric_reconstruction_loss(tgt[sl], s0[sl], s1[sl], d, p0[sl], p1[sl], K[sl], Kinv[sl]) + \
            ol.smooth_loss(d.unsqueeze(1))
        halves.append(l)
    (sum(halves) / 2).backward()
    assert torch.allclose(depth.grad, g_full, rtol=1e-5, atol=1e-9)


def _worker_mixed_schedule(rank, world, port, q):
    """Rank 0 runs the one-stream schedule (L.SERIALIZE, as the kernel-timing pass of bench.py does), rank 1 the overlapped
    one with its communication stream: the ORDER of the bucket exchanges is the order of the collective calls and must be
    identical on both ranks (a mismatch deadlocks or silently mixes buckets).  The second phase
    exchanges with reduce-scatter + all-gather instead of all-reduce: same sums."""
    for p in (ROOT, PKG):
        sys.path.insert(0, p)
    os.environ["MASTER_ADDR"], os.environ["MASTER_PORT"] = "127.0.0.1", str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        import contextlib
        from dvf import lib as L
        from dvf.engine import FlatAdam
        out = {}
        for collective in ("all_reduce", "rs_ag"):
            torch.manual_seed(0)
            params = [torch.nn.Parameter(torch.randn(n)) for n in (3008, 2944, 3072, 1024)]     # (64-float multiples: rs_ag splits evenly)
            opt = FlatAdam(params, lr=1e-3, world_size=world, bucket_mb=0.01, collective=collective)
            assert len(opt.buckets) >= 3
            opt.exchange_log = []
            if rank == 0:
                L.SERIALIZE = True
            else:
                comm, main, cur = (_FakeStream(n) for n in ("comm", "main", "current"))
                opt._comm_stream, opt._main_stream, opt._sides = comm, main, {}
                torch.cuda.current_stream = lambda *a, **k: cur
                torch.cuda.stream = lambda s: contextlib.nullcontext()
            orders = []
            for step in range(3):
                opt.exchange_log.clear()
                opt.zero_grad()
                if rank == 1:
                    opt._main_stream = main
                for p in opt.params:
                    p._dvf_grad.add_(float(rank + 1) * (step + 1))
                    opt.grad_ready(p)
                opt.synchronize_grads()
                orders.append(list(opt.exchange_log))
                for p in params:
                    assert torch.allclose(p.grad, torch.full_like(p, 3.0 * (step + 1))), (collective, rank, step)
            L.SERIALIZE = False
            out[collective] = orders
        gathered = [None, None]
        dist.all_gather_object(gathered, out)
        assert gathered[0] == gathered[1], gathered                       # same bucket order, same collective kind, every step
        for collective, orders in gathered[0].items():
            for step_order in orders:
                assert [i for i, _ in step_order] == sorted(i for i, _ in step_order)          # arena (backward) order
                assert all(kind == collective for _, kind in step_order), step_order
        q.put((rank, "ok"))
    except Exception as e:                                     # noqa: BLE001
        import traceback
        q.put((rank, f"{type(e).__name__}: {e} | " + " / ".join(traceback.format_exc().splitlines()[-4:])))
    finally:
        dist.destroy_process_group()


def test_serialized_and_overlapped_ranks_exchange_buckets_in_the_same_order_world2():
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker_mixed_schedule, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    results = [q.get(timeout=120) for _ in procs]
    for p in procs:
        p.join(timeout=60)
    assert sorted(results) == [(0, "ok"), (1, "ok")], results
