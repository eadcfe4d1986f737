"""Training-step engine: flat parameter / gradient / Adam-moment arenas in HBM, the fused Adam update,
data-parallel gradient buckets (RCCL all-reduce launched from inside backward on a side stream) and HIP-graph
capture of the whole step.

Memory layout (one rank): every parameter of every network lives in ONE contiguous fp32 arena, in the order
gradients become ready during backward (reverse registration order); gradients, exp_avg and exp_avg_sq are
three more arenas with the same offsets.  ``param.data`` / ``param.grad`` are views, so ``state_dict``,
checkpoints and user code see ordinary tensors, while
  * the weight-gradient kernels add straight into the gradient arena (no per-parameter grad tensors),
  * the all-reduce works on contiguous slices of it (no bucket copies),
  * Adam is one kernel launch per contiguous run of parameters that received a gradient.
Reference semantics kept: ``torch.optim.Adam`` as configured at train.py:150-156 / unsupervise.py:241,
including "parameters without a gradient are skipped".
"""
import torch
import torch.distributed as dist

from . import lib as L


class FlatAdam:
    """torch.optim.Adam over flat arenas.  ``params``: iterable of nn.Parameter (all nets together)."""

    def __init__(self, params, lr=1e-3, betas=(0.9, 0.999), eps=1e-8, weight_decay=0.0, world_size=1,
                 bucket_mb=25.0, process_group=None, overlap=True, always_reduce=False, wgrad_stream=True, early_update=True,
                 collective="all_reduce"):
        """overlap=True: buckets are all-reduced on a side stream from inside backward (eager execution).
        overlap=False: backward only marks gradients; ``step()`` all-reduces the buckets on the current stream --
        the mode used when forward+backward are replayed from a HIP graph.  always_reduce: run the exchange even
        with world_size 1 (single-GPU rehearsal of the multi-GPU code path).
        early_update (with overlap): a bucket whose gradients are complete is UPDATED from inside backward on the
        communication / update stream -- (all-reduce,) Adam on the bucket's slice of the arenas, refresh of the packed
        weight copies of its layers -- while the rest of backward runs; ``step()`` only finishes what is left.  The
        deep layers hold most parameters and finish first, so nearly all of the optimizer's HBM streaming hides behind
        the (matrix-bound) backward of the shallow layers.
        collective: how a bucket is exchanged -- "all_reduce" (one RCCL all-reduce: RCCL picks ring / tree / direct), or
        "rs_ag" (reduce-scatter + all-gather in place: the direct all-links form of SURVEY section 8e, every GPU sends 1/world
        of the bucket to each peer over its own xGMI link); the choice only changes which collectives run, not the result
        layout, so an 8-GPU node can A/B the two in one run (bench.py --collective).
        wgrad_stream: weight-gradient kernels run on a second HIP stream.  Nothing in backward consumes a weight
        gradient, so they are off the critical path (dgrad chain) and fill the CUs that a single convolution kernel
        leaves idle; the streams join in ``join_wgrad()`` (called by ``step()``)."""
        self.params = [p for p in params if p.requires_grad][::-1]       # backward-ready order
        if not self.params:
            raise ValueError("no parameters")
        dev = self.params[0].device
        self.lr, self.betas, self.eps, self.weight_decay = float(lr), betas, float(eps), float(weight_decay)
        self.world_size, self.group = int(world_size), process_group
        self.overlap, self.exchange = bool(overlap), (int(world_size) > 1 or always_reduce)
        if collective not in ("all_reduce", "rs_ag"):
            raise ValueError("collective must be 'all_reduce' or 'rs_ag'")
        self.collective = collective
        self.exchange_log = None     # tests: set to a list() to record (bucket index, collective) of every exchange
        sizes = [p.numel() for p in self.params]
        # 64-float (256 B) aligned offsets: every view starts on its own cache lines
        self.offsets, off = [], 0
        for n in sizes:
            self.offsets.append(off)
            off += (n + 63) // 64 * 64
        self.total = off
        self.flat_p = torch.zeros(off, device=dev, dtype=torch.float32)
        self.flat_g = torch.zeros(off, device=dev, dtype=torch.float32)
        self.flat_m = torch.zeros(off, device=dev, dtype=torch.float32)
        self.flat_v = torch.zeros(off, device=dev, dtype=torch.float32)
        self.opt_state = torch.tensor([0.0, self.lr, 0.0, 0.0], device=dev, dtype=torch.float32)
        for p, o, n in zip(self.params, self.offsets, sizes):
            view = self.flat_p[o:o + n].view_as(p)
            view.copy_(p.data)
            p.data = view
            p.grad = None
            p._dvf_grad = self.flat_g[o:o + n].view_as(p)     # wgrad kernels add straight into this
            p._dvf_offset = o
            p._dvf_touched = False
            p._dvf_owner = self
            p._dvf_calls = 0          # grad_ready() calls of the current step ...
            p._dvf_expect = 0         # ... and how many a step makes (learnt from the previous steps)
        # gradient buckets = contiguous arena slices of ~bucket_mb, in backward order
        self.buckets, cur_start, cur_params = [], 0, []
        limit = int(bucket_mb * 1024 * 1024 / 4)
        for i, (p, o, n) in enumerate(zip(self.params, self.offsets, sizes)):
            cur_params.append(p)
            end = o + (n + 63) // 64 * 64
            if end - cur_start >= limit or i == len(self.params) - 1:
                self.buckets.append({"start": cur_start, "end": end, "params": cur_params, "pending": 0})
                cur_start, cur_params = end, []
        for bi, b in enumerate(self.buckets):
            b["index"] = bi
            for p in b["params"]:
                p._dvf_bucket = bi
        self.early_update = bool(early_update) and self.overlap and dev.type == "cuda"
        self._comm_stream = torch.cuda.Stream(device=dev) if ((self.exchange or self.early_update) and self.overlap and
                                                               dev.type == "cuda") else None
        self._adam_started = False   # the step counter has been advanced for the current step
        self._main_stream = None     # compute stream of the step (recorded by zero_grad)
        self.n_reduced = 0           # buckets exchanged so far (tests / bench bookkeeping)
        self._ranges = None
        self._stable = False
        self._reset_pending()
        self.side = torch.cuda.Stream(device=dev) if (wgrad_stream and dev.type == "cuda") else None
        self._sides = {}         # compute stream -> its weight-gradient stream (the main stream's is self.side)
        self._fork_cache = {}    # raw stream handle -> [stream object, side stream, event ring, next slot]
        self._side_dirty = False
        self._keep = []          # tensors read by side-stream kernels, kept alive until the join

    # ------------------------------------------------------------------ second stream for weight gradients
    def fork_wgrad(self, *tensors):
        """Make the side stream wait for everything enqueued so far on the current stream; returns the side stream
        (or None: run on the current stream).  ``tensors`` are kept alive until ``join_wgrad``."""
        if self.side is None or L.SERIALIZE:
            return None
        # (host cost matters: this runs once per convolution of the backward pass.  The raw stream handle keys a cache of
        # (stream object, side stream, ring of reusable events): a wait captures the event's state when it is enqueued, so
        # re-recording a ring slot later does not disturb an earlier wait.)
        raw = L.stream_raw()
        ent = self._fork_cache.get(raw)
        if ent is None:
            cur = torch.cuda.current_stream()
            side = self._sides.get(cur.cuda_stream)
            if side is None:     # one per compute stream: a sub-network on the auxiliary stream gets its own
                side = self._sides[cur.cuda_stream] = self.side if not self._sides else torch.cuda.Stream(device=cur.device)
            side._dvf_raw = side.cuda_stream      # (ConvFn.backward launches on the side stream by handle)
            ent = self._fork_cache[raw] = [cur, side, [torch.cuda.Event() for _ in range(32)], 0]
        cur, side, ring, i = ent
        ent[3] = (i + 1) & 31
        ev = ring[i]
        ev.record(cur)
        side.wait_event(ev)
        self._keep.extend(t for t in tensors if t is not None)
        self._side_dirty = True
        return side

    def join_wgrad(self):
        if self._side_dirty:
            for side in self._sides.values():
                torch.cuda.current_stream().wait_stream(side)
            self._side_dirty = False
        self._keep.clear()

    # ------------------------------------------------------------------ autograd side (called from ConvFn.backward)
    def _reset_pending(self):
        """Start of a step.  A bucket may leave from inside backward only when the LAST gradient contribution of every
        parameter in it has been enqueued.  A parameter receives one grad_ready() call per use of its module in the step
        (a network applied to two images calls twice: shared weights), so the engine learns the number of calls per
        parameter from the step before and launches a bucket when every touched parameter has reached its count.  The
        first step (counts unknown) runs without early launches: ``step()`` exchanges and updates everything.  A step that
        makes FEWER calls than the previous one simply leaves the affected buckets to ``step()``; one that makes MORE is
        caught in grad_ready()."""
        stable = self._ranges is not None
        for p in self.params:
            if p._dvf_expect < 0:     # relearn(): one step without early launches
                stable = False
            p._dvf_expect = p._dvf_calls
            p._dvf_calls = 0
        self._stable = stable
        for b in self.buckets:
            b["pending"] = sum(p._dvf_expect for p in b["params"]) if stable else -1      # -1: no early launch this step
            b["launched"] = False
            b["updated"] = False
        self._adam_started = False

    def grad_ready(self, p):
        """A weight/bias gradient contribution has been enqueued on the compute stream."""
        if not p._dvf_touched:
            p._dvf_touched = True
            p.grad = p._dvf_grad            # expose it the torch way
            self._ranges = None
        p._dvf_calls += 1
        if not ((self.exchange or self.early_update) and self.overlap):
            return
        b = self.buckets[p._dvf_bucket]
        if b["launched"]:
            # its bucket has already been exchanged / updated: this contribution would be lost (and the remaining dgrads
            # of the step would read updated weights).  Only possible when the step's graph changed under the engine.
            raise RuntimeError("FlatAdam: a gradient of a parameter arrived after its bucket had left (the step uses a module "
                               "more often than the previous step did); run this step with early launches disabled "
                               "(FlatAdam.relearn()) or construct the optimizer with overlap=False")
        if b["pending"] < 0 or p._dvf_calls > p._dvf_expect:
            b["pending"] = -1               # unexpected call pattern: leave this bucket to step()
            return
        b["pending"] -= 1
        if b["pending"] == 0 and self._ranges is not None:
            self._launch_bucket(b)

    def relearn(self):
        """Forget the learnt call counts: the next step runs without early launches and re-learns them (call this when
        the step function changes, e.g. a network starts being applied a second time)."""
        for p in self.params:
            p._dvf_expect = -1

    def _bucket_ranges(self, b):
        """Contiguous arena ranges of the bucket's parameters that received a gradient."""
        rngs = []
        for p in b["params"]:
            if not p._dvf_touched:
                continue
            o = p._dvf_offset
            end = o + (p.numel() + 63) // 64 * 64
            if rngs and rngs[-1][1] == o:
                rngs[-1][1] = end
            else:
                rngs.append([o, end])
        return rngs

    def _adam_bucket(self, b):
        """Fused Adam on the bucket's slice of the arenas + refresh of the packed weight copies of its layers, on the
        current stream."""
        lib = L.lib()
        for o, e in self._bucket_ranges(b):
            L.check(lib.dvf_adam_step(L.dev(self.flat_p[o:e]), L.dev(self.flat_g[o:e]), L.dev(self.flat_m[o:e]),
                                      L.dev(self.flat_v[o:e]), e - o, L.dev(self.opt_state), 0 if self._adam_started else 1,
                                      self.betas[0], self.betas[1], self.eps, self.weight_decay,
                                      1.0 / self.world_size, L.stream()), "dvf_adam_step")
            self._adam_started = True
        for p in b["params"]:            # packed convolution weights (dvf/conv.py) of the updated layers are stale now
            if p._dvf_touched:
                p._dvf_epoch = getattr(p, "_dvf_epoch", 0) + 1
        from . import conv as _conv
        _conv.repack_all(owner=self, bucket=b["index"])
        b["updated"] = True

    def _exchange(self, b, grads):
        """Sum of the bucket's gradient slice over the ranks, in place."""
        world = dist.get_world_size(self.group) if dist.is_initialized() else 1
        n = grads.numel()
        if self.collective == "rs_ag" and world > 1 and n % world == 0:
            shard = grads[(n // world) * dist.get_rank(self.group):(n // world) * (dist.get_rank(self.group) + 1)]
            dist.reduce_scatter_tensor(shard, grads, op=dist.ReduceOp.SUM, group=self.group)
            dist.all_gather_into_tensor(grads, shard, group=self.group)
            kind = "rs_ag"
        else:
            dist.all_reduce(grads, op=dist.ReduceOp.SUM, group=self.group)
            kind = "all_reduce"
        self.n_reduced += 1
        if self.exchange_log is not None:
            self.exchange_log.append((b["index"], kind))

    def _launch_bucket(self, b):
        b["launched"] = True
        grads = self.flat_g[b["start"]:b["end"]]
        if self._comm_stream is None or L.SERIALIZE:        # CPU tensors (gloo), overlap=False or serialised: current stream
            if self.exchange:
                self._exchange(b, grads)
            return
        # The bucket's gradients were written by kernels on EVERY compute stream of the step: bias gradients and thin
        # layers on the stream the layer's backward ran on (the main stream for the depth network, the auxiliary stream
        # for the pose network), weight gradients on the side stream behind each of the two.  The exchange / update
        # waits for all of them (and, for the update, for the dgrad kernels that still read the bucket's weights: they
        # were enqueued on those same streams before this point).
        comm = self._comm_stream
        comm.wait_stream(torch.cuda.current_stream())         # the stream this grad_ready() was called on
        if self._main_stream is not None:
            comm.wait_stream(self._main_stream)               # the stream zero_grad()/backward() were issued from
        for side in self._sides.values():
            comm.wait_stream(side)
        for aux in L.aux_streams_on(comm.device):
            comm.wait_stream(aux)
        with torch.cuda.stream(comm):
            if self.exchange:
                self._exchange(b, grads)
            if self.early_update and self._ranges is not None:
                self._adam_bucket(b)

    # ------------------------------------------------------------------ optimizer API
    def zero_grad(self):
        if self.flat_g.is_cuda:
            self._main_stream = torch.cuda.current_stream()
        L.join_aux_streams()
        self.join_wgrad()
        self.flat_g.zero_()
        self._reset_pending()

    def _touched_ranges(self):
        if self._ranges is None:
            rngs = []
            for p, o in zip(self.params, self.offsets):
                if not p._dvf_touched:
                    continue
                end = o + (p.numel() + 63) // 64 * 64
                if rngs and rngs[-1][1] == o:
                    rngs[-1][1] = end
                else:
                    rngs.append([o, end])
            self._ranges = rngs
        return self._ranges

    def synchronize_grads(self):
        """Finish the data-parallel exchange (and the early updates): launch what backward could not (first step,
        stragglers) and make the compute stream wait for the communication / update stream."""
        self._touched_ranges()
        if self.exchange:
            for b in self.buckets:
                if not b["launched"] and any(p._dvf_touched for p in b["params"]):
                    self._launch_bucket(b)
        if self._comm_stream is not None:
            torch.cuda.current_stream().wait_stream(self._comm_stream)

    def step(self):
        L.join_aux_streams()         # backward kernels of a sub-network that ran on the auxiliary stream
        self.join_wgrad()
        self.synchronize_grads()
        for b in self.buckets:       # buckets that were not updated from inside backward
            if not b["updated"] and any(p._dvf_touched for p in b["params"]):
                self._adam_bucket(b)

    def set_lr(self, lr):
        self.lr = float(lr)
        self.opt_state[1:2].fill_(self.lr)

    def state_dict(self):
        return {"opt_state": self.opt_state.clone(), "exp_avg": self.flat_m.clone(), "exp_avg_sq": self.flat_v.clone(),
                "lr": self.lr, "betas": self.betas, "eps": self.eps, "weight_decay": self.weight_decay}

    def load_state_dict(self, sd):
        self.opt_state.copy_(sd["opt_state"])
        self.flat_m.copy_(sd["exp_avg"])
        self.flat_v.copy_(sd["exp_avg_sq"])


class GraphedStep:
    """Capture ``fn(*static_inputs)`` -- a full forward + backward + optimizer step -- into a HIP graph after
    ``warmup`` eager runs on a side stream, then replay it.  ``fn`` must return a tuple of tensors (losses);
    new batches are copied into the static input tensors before each replay."""

    def __init__(self, fn, static_inputs, warmup=2):
        self.fn, self.inputs = fn, list(static_inputs)
        s = torch.cuda.Stream()
        s.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(s):
            for _ in range(warmup):
                fn(*self.inputs)
        torch.cuda.current_stream().wait_stream(s)
        torch.cuda.synchronize()
        self.graph = torch.cuda.CUDAGraph()
        with torch.cuda.graph(self.graph):
            self.outputs = fn(*self.inputs)
        self.eager_steps = warmup

    def __call__(self, *new_inputs):
        for dst, src in zip(self.inputs, new_inputs):
            if src is not None and src is not dst:
                dst.copy_(src, non_blocking=True)
        self.graph.replay()
        # The replay updated the weights through raw pointers and refreshed the packed copies that existed at capture time;
        # no Python bookkeeping ran.  Bump the manual epoch so that every packed copy is re-validated (and repacked) the next
        # time EAGER code uses it -- e.g. validate() between replays with a batch size the capture never saw: its packed
        # copy would otherwise keep a matching stamp and serve the weights of the epoch it was first built in.
        L.PACK_EPOCH += 1
        return self.outputs
