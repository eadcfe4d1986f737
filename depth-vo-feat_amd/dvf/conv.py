"""Convolution / transposed convolution with fused bias + activation as autograd Functions over the
fp32-MFMA kernels of libdvf_hip.so, the small memory-bound ops around them, and the nn.Module shells whose
parameter names and shapes equal the reference's (so ``state_dict`` files interchange)."""
import ctypes
import math
import weakref

import torch
import torch.nn as nn

from . import lib as L


def _c(t):
    return t if (t.dtype == torch.float32 and t.is_contiguous()) else t.float().contiguous()


class _NullCtx:
    def __enter__(self):
        return self

    def __exit__(self, *exc):
        return False


_NULL = _NullCtx()
_NULL_PTRS = [(ctypes.c_void_p * max(n, 1))() for n in range(9)]    # all-NULL pointer tables by length


def conv_out_size(h, k, stride, pad, opad, transposed):
    return (h - 1) * stride - 2 * pad + k + opad if transposed else (h + 2 * pad - k) // stride + 1


_WS = {}      # (device, stream) -> split-K scratch shared by the convolutions of that stream


def _workspace(nfloats, device, raw_stream=None):
    """Scratch for split-K partial tiles: one buffer per compute stream (kernels of one stream are ordered, so they can
    share it); it only grows.  During a HIP-graph capture the current stream is the capture's own, so there is no
    cached buffer for it: the scratch then comes from the capture's private memory pool (one allocation per call, at
    capture time only) -- the captured step must run the same split-K reduction as the eager one, not the float-atomic
    fallback of a missing workspace."""
    key = (device, L.stream_raw() if raw_stream is None else raw_stream)
    buf = _WS.get(key)
    if buf is None or buf.numel() < nfloats:
        if torch.cuda.is_current_stream_capturing():
            return torch.empty(max(int(nfloats), 1), device=device, dtype=torch.float32)
        _WS[key] = buf = torch.empty(max(int(nfloats), 1 << 20), device=device, dtype=torch.float32)
    return buf


_PACK_REGISTRY = []    # [weakref(weight), desc, segc, kind, cache entry] of every packed copy in use
_PACK_TABLES = {}      # (id(owner), bucket) -> job tables of the layers of one optimizer bucket (None: all layers)
_PACK_SMALL_LDS = 20 * 1024    # bytes: eight 256-thread pack blocks of this size share a CU


def _stamp(w):
    """What a packed copy was made from: storage, torch's version counter, the owning optimizer's update count of this
    parameter (FlatAdam updates through raw pointers) and the manual invalidation epoch."""
    return (w.data_ptr(), w._version, getattr(w, "_dvf_epoch", 0), L.PACK_EPOCH)


def repack_all(owner=None, bucket=None):
    """Refresh packed weight copies with one launch per tile-size group (called by FlatAdam right after an update, so
    the convolutions of the next step find their copies current).  owner/bucket: only the layers whose weight lives in
    that bucket of that optimizer (torch.optim users call ``repack_all()`` for everything, or rely on the lazy
    per-layer refresh driven by the weight's version counter).  The job tables are built once per set of layers."""
    if not _PACK_REGISTRY:
        return
    lib = L.lib()
    live = [(r, r[0]()) for r in _PACK_REGISTRY]
    live = [(r, w) for r, w in live if w is not None]
    if len(live) != len(_PACK_REGISTRY) and not torch.cuda.is_current_stream_capturing():
        _PACK_REGISTRY[:] = [r for r, _ in live]          # drop the layers of models that no longer exist
    if owner is not None:
        live = [(r, w) for r, w in live if getattr(w, "_dvf_owner", None) is owner and w._dvf_bucket == bucket]
    if not live:
        return
    t = _PACK_TABLES.setdefault((id(owner) if owner is not None else 0, bucket), {"n": -1})
    key = tuple((id(r), w.data_ptr()) for r, w in live)
    if t.get("key") != key:
        if torch.cuda.is_current_stream_capturing():
            # the set of live layers changed since the table was built (e.g. another model was garbage-collected) and a
            # table cannot be uploaded inside a capture: record one pack launch per layer instead -- a captured step
            # must refresh the copies on every replay
            for r, w in live:
                _, desc, segc, kind, ent = r
                L.check(lib.dvf_conv2d_pack(ctypes.byref(desc), L.int_array(segc), len(segc), kind, L.dev(w), L.dev(ent[0]),
                                            L.stream()), "dvf_conv2d_pack")
                ent[1] = _stamp(w)
            return
        # jobs are grouped by the LDS tile they need: a launch reserves the maximum of its jobs for every block, and
        # the few large-kernel layers (7x7, 5x5: ~50 KB) would otherwise hold the 3x3 bulk to three blocks per CU
        groups = {}
        for r, w in live:
            _, desc, segc, kind, ent = r
            buf = (ctypes.c_char * (L.PACK_JOB_BYTES * len(segc)))()
            nb = (ctypes.c_int * len(segc))()
            lds = ctypes.c_int(4)
            n = lib.dvf_conv2d_pack_jobs(ctypes.byref(desc), L.int_array(segc), len(segc), kind, L.dev(w), L.dev(ent[0]),
                                         ctypes.cast(buf, ctypes.c_void_p), len(segc), nb, ctypes.byref(lds))
            if n < 0:
                L.check(n, "dvf_conv2d_pack_jobs")
            g = groups.setdefault(0 if lds.value <= _PACK_SMALL_LDS else 1, {"blobs": [], "blocks": [], "lds": 4})
            g["blobs"].append(bytes(buf)[: n * L.PACK_JOB_BYTES])
            g["blocks"] += list(nb[:n])
            g["lds"] = max(g["lds"], lds.value)
        dev = live[0][1].device
        # (earlier tables stay allocated: a captured graph may still launch with them)
        t.setdefault("keep", []).append(t.get("launches"))
        t["launches"] = []
        for _, g in sorted(groups.items()):
            blocks = g["blocks"]
            prefix = [0]
            for b in blocks:
                prefix.append(prefix[-1] + b)
            t["launches"].append({
                "jobs": torch.frombuffer(bytearray(b"".join(g["blobs"])), dtype=torch.uint8).to(dev),
                "prefix": torch.tensor(prefix, dtype=torch.int32, device=dev),
                "block_job": torch.repeat_interleave(torch.arange(len(blocks), dtype=torch.int32),
                                                     torch.tensor(blocks, dtype=torch.int64)).to(dev),
                "njobs": len(blocks), "total": prefix[-1], "lds": g["lds"]})
        t["key"] = key
    for q in t["launches"]:
        L.check(lib.dvf_conv2d_pack_batch(q["jobs"].data_ptr(), q["prefix"].data_ptr(), q["block_job"].data_ptr(), q["njobs"],
                                          q["total"], q["lds"], L.stream()), "dvf_conv2d_pack_batch")
    for r, w in live:
        r[4][1] = _stamp(w)


def _packed_weights(weight, holder, desc, segc, kind, pkey=None, segarr=None):
    """Packed copy of `weight` for the pipelined kernels (kind 0: forward, 1: dgrad), cached on `holder` (the
    parameter) per layer geometry and refreshed when the weights changed.  None when the plan is unsupported."""
    cache = holder.__dict__.get("_dvf_pack")
    if cache is None:
        cache = holder.__dict__["_dvf_pack"] = {}
    if pkey is None:
        pkey = (desc.N, desc.H_in, desc.W_in, desc.H_out, desc.W_out, desc.stride, desc.pad, desc.transposed, tuple(segc))
    key = (kind, pkey)
    ent = cache.get(key)
    if ent is None:
        lib = L.lib()
        if segarr is None:
            segarr = L.int_array(segc)
        nf = lib.dvf_conv2d_packed_floats(ctypes.byref(desc), segarr, len(segc), kind)
        # split-K scratch of this op (packed and unpacked kernels alike): partial tiles are reduced in a fixed order
        wsf = max(int(lib.dvf_conv2d_ws_floats(ctypes.byref(desc), segarr, len(segc), kind)), 0)
        if nf == L.ERR_UNSUPPORTED:
            cache[key] = ent = [None, None, wsf]
        else:
            if nf < 0:
                L.check(int(nf), "dvf_conv2d_packed_floats")
            # zero-filled once: padding slots of the packed image are never written again
            cache[key] = ent = [torch.zeros(int(nf), device=weight.device, dtype=torch.float32), None, wsf]
            _PACK_REGISTRY.append([weakref.ref(holder), desc, list(segc), kind, ent])
    buf = ent[0]
    if buf is None:
        return None, (_workspace(ent[2], weight.device) if ent[2] else None)
    stamp = (holder.data_ptr(), holder._version, getattr(holder, "_dvf_epoch", 0), L.PACK_EPOCH)     # == _stamp(holder)
    if ent[1] != stamp:
        with L.timed("conv_pack", 0.0, 8.0 * weight.numel()):
            L.check(L.lib().dvf_conv2d_pack(ctypes.byref(desc), L.int_array(segc), len(segc), kind, L.dev(weight, "weight"),
                                            L.dev(buf), L.stream()), "dvf_conv2d_pack")
        ent[1] = stamp
    return buf, (_workspace(ent[2], weight.device) if ent[2] else None)


DETERMINISTIC = False   # True: weight / bias gradients are summed in a fixed order (scratch + ordered finish) instead of with
                        # float atomics in order of arrival: bit-identical gradient arenas from identical state, as on the
                        # reference's CPU path, for ~5 % of a cfg-2 step (set_deterministic(); entry scripts: --deterministic)


def set_deterministic(on=True):
    global DETERMINISTIC
    DETERMINISTIC = bool(on)


def _bias_sums(lib, dy, y, dpre, dbias, N, C, HW, act, alpha, beta, accumulate):
    """dvf_act_bwd2 (dpre = dy * act'(y), dbias (+)= channel sums), through the deterministic entry when asked."""
    if DETERMINISTIC and dbias is not None:
        ws = _workspace(int(lib.dvf_act_bwd_ws_floats(N, C, HW)), dy.device)
        L.check(lib.dvf_act_bwd_det(L.dev(dy, "grad_out"), L.dev(y), L.dev(dpre), L.dev(dbias), N, C, HW, act, alpha, beta,
                                    1 if accumulate else 0, L.dev(ws), ws.numel(), L.stream()), "dvf_act_bwd_det")
    else:
        L.check(lib.dvf_act_bwd2(L.dev(dy, "grad_out"), L.dev(y), L.dev(dpre), L.dev(dbias), N, C, HW, act, alpha, beta,
                                 1 if accumulate else 0, L.stream()), "dvf_act_bwd2")


def _wgrad_ws_floats(holder, desc, segc):
    """Scratch floats of the deterministic weight-gradient flush for this layer geometry (cached on the parameter)."""
    cache = holder.__dict__.setdefault("_dvf_wgws", {})
    key = (desc.N, desc.H_in, desc.W_in, desc.H_out, desc.W_out, desc.stride, desc.pad, desc.transposed, tuple(segc))
    n = cache.get(key)
    if n is None:
        n = int(L.lib().dvf_conv2d_wgrad_ws_floats(ctypes.byref(desc), L.int_array(segc), len(segc)))
        cache[key] = n = max(n, 0)
    return n


GEOM_LOG = None   # set to a list(): ConvFn.forward appends the geometry of every call (tests: bench-shape parity)
FUSE_RELU_BWD = True   # False: every layer runs its own activation backward pass (dvf_act_bwd2), as in round 1


class ReluTag:
    """Carried by the output y of a ReLU convolution whose module was declared ``fuse_bwd`` (EVERY consumer of y is a
    ConvFn): the consumers' dgrad kernels then deliver dL/dy already multiplied by relu'(y) -- the producing layer runs no
    activation backward pass, and its bias gradient is one more column of its weight-gradient GEMM (dvf_conv2d_wgrad_bias).  relu' is a 0/1
    mask, so masking each consumer's contribution equals masking their sum."""
    __slots__ = ()


class _Geo:
    """One layer geometry: descriptor, segment table and labels (built once per (cfg, N, H, W, segment channels))."""
    __slots__ = ("desc", "dref", "segc", "segarr", "nseg", "cin0", "cout", "oh", "ow", "macs", "tag", "pkey")

    def __init__(self, weight, cfg, inputs):
        k, stride, pad, opad, transposed, act, alpha, beta, out_hw = cfg[:9]
        N, _, H, W = inputs[0].shape
        self.segc = segc = [int(x.shape[1]) for x in inputs]
        self.nseg, self.cin0 = len(segc), segc[0]
        self.segarr = L.int_array(segc)
        cin = sum(segc)
        self.cout = cout = weight.shape[1] if transposed else weight.shape[0]
        if (weight.shape[0] if transposed else weight.shape[1]) != cin:
            raise ValueError(f"weight {tuple(weight.shape)} does not match {cin} input channels")
        oh, ow = conv_out_size(H, k, stride, pad, opad, transposed), conv_out_size(W, k, stride, pad, opad, transposed)
        if out_hw is not None:                              # crop_like folded into the kernel
            oh, ow = min(oh, out_hw[0]), min(ow, out_hw[1])
        self.oh, self.ow = oh, ow
        self.desc = L.ConvDesc(N, cin, H, W, cout, oh, ow, k, k, stride, pad, 1 if transposed else 0, act, alpha, beta)
        self.dref = ctypes.byref(self.desc)
        # algorithmic MACs of this layer: kept pixels x taps actually contributing
        taps = k * k / (stride * stride) if transposed else k * k
        self.macs = float(N) * cout * oh * ow * cin * taps
        self.tag = f"{'T' if transposed else 'C'}{k}x{k}s{stride} {cin}->{cout} in{H}x{W} out{oh}x{ow} N{N}"
        self.pkey = (N, H, W, oh, ow, stride, pad, 1 if transposed else 0, tuple(segc))


class ConvFn(torch.autograd.Function):
    """act(conv(cat(inputs), weight) + bias).  cfg = (k, stride, pad, opad, transposed, act, alpha, beta, out_hw)."""

    @staticmethod
    def forward(ctx, weight, bias, cfg, *inputs):
        k, stride, pad, opad, transposed, act, alpha, beta, out_hw = cfg[:9]
        ctx.out_tag = cfg[9] if len(cfg) > 9 else None      # ReluTag of this layer's own output (module declared fuse_bwd)
        ctx.in_tags = [getattr(x, "_dvf_relu_tag", None) for x in inputs]
        # parameters owned by a FlatAdam arena: their gradients are added in place (dvf/engine.py)
        ctx.wparam = weight if getattr(weight, "_dvf_grad", None) is not None else None
        ctx.bparam = bias if (bias is not None and getattr(bias, "_dvf_grad", None) is not None) else None
        ctx.holder = weight
        inputs = [_c(x) for x in inputs]
        weight = _c(weight)
        bias = _c(bias) if bias is not None else None
        x0 = inputs[0]
        N, _, H, W = x0.shape
        # everything that depends on the layer geometry alone (descriptor, sizes, labels) is made once per geometry and kept
        # on the parameter: this function runs ~50 times per step and the host is the second bound of the step
        gkey = (cfg[:9], N, H, W) if len(inputs) == 1 else (cfg[:9], N, H, W) + tuple(x.shape[1] for x in inputs[1:])
        geos = ctx.holder.__dict__.get("_dvf_geo")
        if geos is None:
            geos = ctx.holder.__dict__["_dvf_geo"] = {}
        g = geos.get(gkey)
        if g is None:
            g = geos[gkey] = _Geo(weight, cfg, inputs)
        elif g.cin0 != x0.shape[1]:
            raise ValueError(f"weight {tuple(weight.shape)} does not match {sum(int(x.shape[1]) for x in inputs)} input channels")
        if g.nseg > 1:
            for x in inputs:
                if x.shape[0] != N or x.shape[2] != H or x.shape[3] != W:
                    raise ValueError(f"virtual concat needs equal N,H,W; got {[tuple(t.shape) for t in inputs]}")
        desc, segc, cout, oh, ow = g.desc, g.segc, g.cout, g.oh, g.ow
        if GEOM_LOG is not None:
            GEOM_LOG.append((tuple(segc), cout, tuple(cfg[:9]), (N, H, W), tuple(ctx.needs_input_grad[3:])))
        out = torch.empty((N, cout, oh, ow), device=weight.device, dtype=torch.float32)
        ctx.macs, ctx.tag, ctx.geo = g.macs, g.tag, g
        packed, ws = _packed_weights(weight, ctx.holder, desc, segc, 0, g.pkey, g.segarr)
        # (operands went through _c(): float32 and contiguous; a CPU tensor is refused here, there is no CPU path)
        if not x0.is_cuda:
            L.dev(x0, "input")
        inarr = (ctypes.c_void_p * g.nseg)(*[x.data_ptr() for x in inputs])
        wsp, wsn = (ws.data_ptr(), ws.numel()) if ws is not None else (None, 0)
        bp = bias.data_ptr() if bias is not None else None
        lib = L.lib()
        with (L.timed("conv_fwd", 2 * g.macs, tag=g.tag) if L.TIMER is not None else _NULL):
            rc = L.ERR_UNSUPPORTED
            if packed is not None:
                rc = lib.dvf_conv2d_fwd_packed(g.dref, inarr, g.segarr, g.nseg, packed.data_ptr(), bp, out.data_ptr(),
                                               wsp, wsn, L.stream_raw())
                if rc != 0 and rc != L.ERR_UNSUPPORTED:     # (unsupported at run time: an operand is not 16-byte aligned)
                    L.check(rc, "dvf_conv2d_fwd_packed")
            if rc == L.ERR_UNSUPPORTED:
                rc = lib.dvf_conv2d_fwd_ws(g.dref, inarr, g.segarr, g.nseg, weight.data_ptr(), bp, out.data_ptr(), wsp, wsn,
                                           L.stream_raw())
                if rc != 0:
                    L.check(rc, "dvf_conv2d_fwd_ws")
        if L.PLAN_LOG is not None:
            L.note_plans("fwd")
        ctx.save_for_backward(weight, out, *inputs)
        ctx.desc, ctx.segc, ctx.has_bias, ctx.inarr = desc, segc, bias is not None, inarr
        return out

    @staticmethod
    def backward(ctx, gout):
        weight, out, *inputs = ctx.saved_tensors
        desc, segc, g = ctx.desc, ctx.segc, ctx.geo
        lib = L.lib()
        gout = _c(gout)
        N, cout, oh, ow = out.shape
        nig = ctx.needs_input_grad
        need_w, need_b = nig[0], ctx.has_bias and nig[1]
        need_in = nig[3:]
        timing = L.TIMER is not None
        st = L.stream_raw()
        # a bias owned by a FlatAdam arena: its gradient slice is zero after zero_grad(), add the channel sums in place
        acc_b = need_b and ctx.bparam is not None
        dbias = (ctx.bparam._dvf_grad if acc_b else torch.empty(cout, device=out.device)) if need_b else None
        fused_out = ctx.out_tag is not None
        if fused_out:
            # every consumer's dgrad already applied relu'(y) and accumulated the bias gradient into the tag's buffer
            # (the bias gradient rides on the weight-gradient kernel below: one more column, of ones)
            dpre = gout
        elif desc.act != L.ACT_NONE:
            dpre = torch.empty_like(gout)
            with (L.timed("act_bwd", 0.0, 12.0 * gout.numel()) if timing else _NULL):
                _bias_sums(lib, gout, out, dpre, dbias, N, cout, oh * ow, desc.act, desc.alpha, desc.beta, acc_b)
        else:
            dpre = gout
            if need_b:
                _bias_sums(lib, gout, None, None, dbias, N, cout, oh * ow, L.ACT_NONE, 1.0, 0.0, acc_b)
        if not dpre.is_cuda:
            L.dev(dpre, "grad_out")
        dprep = dpre.data_ptr()
        any_in = True in need_in
        gins = [torch.empty_like(x) if need else None for x, need in zip(inputs, need_in)] if any_in else [None] * g.nseg
        if any_in:
            packed, ws = _packed_weights(weight, ctx.holder, desc, segc, 1, g.pkey, g.segarr)
            wsp, wsn = (ws.data_ptr(), ws.numel()) if ws is not None else (None, 0)
            pp = packed.data_ptr() if packed is not None else None
            ginarr = (ctypes.c_void_p * g.nseg)(*[t.data_ptr() if t is not None else None for t in gins])
            masked = [t is not None and need for t, need in zip(ctx.in_tags, need_in)]
            if timing:
                frac = sum(c for c, need in zip(segc, need_in) if need) / float(sum(segc))
            with (L.timed("conv_dgrad", 2 * g.macs * frac, tag=g.tag) if timing else _NULL):
                if True in masked:
                    # segments produced by fuse_bwd ReLU layers: their gradient leaves masked, their bias gradient summed
                    # (a segment whose packed plan is refused at run time runs unpacked inside the library)
                    marr = (ctypes.c_void_p * g.nseg)(*[x.data_ptr() if m else None for x, m in zip(inputs, masked)])
                    rc = lib.dvf_conv2d_dgrad_masked(g.dref, dprep, pp, weight.data_ptr(), ginarr, g.segarr, g.nseg, wsp, wsn,
                                                     marr, _NULL_PTRS[g.nseg], st)
                    if rc != 0:
                        L.check(rc, "dvf_conv2d_dgrad_masked")
                else:
                    rc = L.ERR_UNSUPPORTED
                    if packed is not None:
                        rc = lib.dvf_conv2d_dgrad_packed(g.dref, dprep, pp, weight.data_ptr(), ginarr, g.segarr, g.nseg, wsp, wsn, st)
                        if rc != 0 and rc != L.ERR_UNSUPPORTED:
                            L.check(rc, "dvf_conv2d_dgrad_packed")
                    if rc == L.ERR_UNSUPPORTED:
                        rc = lib.dvf_conv2d_dgrad_ws(g.dref, dprep, weight.data_ptr(), ginarr, g.segarr, g.nseg, wsp, wsn, st)
                        if rc != 0:
                            L.check(rc, "dvf_conv2d_dgrad_ws")
            if L.PLAN_LOG is not None:
                L.note_plans("dgrad")
        dw = None
        if fused_out and need_b and not need_w:     # (frozen weights, trainable bias: one reduction pass over dpre)
            _bias_sums(lib, dpre, None, None, dbias, N, cout, oh * ow, L.ACT_NONE, 1.0, 0.0, acc_b)
        if need_w:
            arena = ctx.wparam is not None
            dw = ctx.wparam._dvf_grad if arena else torch.empty_like(weight)
            # with a FlatAdam arena the weight gradient is not consumed inside backward: it goes to the side stream.  The
            # launch names that stream by its handle (switching torch's current stream costs ~20 us of host time per layer);
            # only the timing passes, whose events are recorded on the current stream, switch it.
            side = ctx.wparam._dvf_owner.fork_wgrad(dpre, *inputs) if arena else None
            wst = side._dvf_raw if side is not None else st
            with (torch.cuda.stream(side) if (side is not None and timing) else _NULL):
                with (L.timed("conv_wgrad", 2 * g.macs, tag=g.tag) if timing else _NULL):
                    # (DETERMINISTIC: partial tiles through a per-stream scratch, summed in fixed order; else float atomics)
                    wsf = _wgrad_ws_floats(ctx.wparam if arena else weight, desc, segc) if DETERMINISTIC else 0
                    wws = _workspace(wsf, dpre.device, wst) if wsf > 0 else None
                    rc = lib.dvf_conv2d_wgrad_det(g.dref, ctx.inarr, g.segarr, g.nseg, dprep, dw.data_ptr(), 1 if arena else 0,
                                                  dbias.data_ptr() if (fused_out and need_b) else None, 1 if acc_b else 0,
                                                  wws.data_ptr() if wws is not None else None,
                                                  wws.numel() if wws is not None else 0, wst)
                    if rc != 0:
                        L.check(rc, "dvf_conv2d_wgrad_det")
                    if L.PLAN_LOG is not None:
                        L.note_plans("wgrad")
            if arena:
                ctx.wparam._dvf_owner.grad_ready(ctx.wparam)
                dw = None
        if acc_b:
            ctx.bparam._dvf_owner.grad_ready(ctx.bparam)
            dbias = None
        return (dw, dbias, None, *gins)


class Upsample2xFn(torch.autograd.Function):
    """F.interpolate(x, scale_factor=2, mode='bilinear', align_corners=False) followed by crop_like to out_hw
    (reference DispNetS.py:115,121,127)."""

    @staticmethod
    def forward(ctx, x, out_hw):
        x = _c(x)
        N, C, H, W = x.shape
        oh, ow = min(2 * H, out_hw[0]), min(2 * W, out_hw[1])
        out = torch.empty((N, C, oh, ow), device=x.device, dtype=torch.float32)
        L.check(L.lib().dvf_resize_bilinear_fwd(L.dev(x, "input"), L.dev(out), N * C, H, W, oh, ow, 0.5, 0.5, L.stream()),
                "dvf_resize_bilinear_fwd")
        ctx.shape = (N, C, H, W, oh, ow)
        return out

    @staticmethod
    def backward(ctx, gout):
        N, C, H, W, oh, ow = ctx.shape
        gin = torch.empty((N, C, H, W), device=gout.device, dtype=torch.float32)
        L.check(L.lib().dvf_upsample2x_bwd(L.dev(_c(gout), "grad_out"), L.dev(gin), N * C, H, W, oh, ow, L.stream()),
                "dvf_upsample2x_bwd")
        return gin, None


class RecipFn(torch.autograd.Function):
    """1 / (x + eps): disparity -> depth (train.py:188 with eps=0, unsupervise.py:99 with eps=1e-4)."""

    @staticmethod
    def forward(ctx, x, eps):
        x = _c(x)
        y = torch.empty_like(x)
        L.check(L.lib().dvf_recip_fwd(L.dev(x, "input"), L.dev(y), eps, x.numel(), L.stream()), "dvf_recip_fwd")
        ctx.save_for_backward(y)
        return y

    @staticmethod
    def backward(ctx, gy):
        (y,) = ctx.saved_tensors
        gx = torch.empty_like(y)
        L.check(L.lib().dvf_recip_bwd(L.dev(_c(gy), "grad_out"), L.dev(y), L.dev(gx), y.numel(), L.stream()),
                "dvf_recip_bwd")
        return gx, None


class SpatialMeanFn(torch.autograd.Function):
    """scale * x.mean(3).mean(2)  (PoseExpNet_sfm.py:72-73)."""

    @staticmethod
    def forward(ctx, x, scale):
        x = _c(x)
        N, C, H, W = x.shape
        out = torch.empty((N, C), device=x.device, dtype=torch.float32)
        L.check(L.lib().dvf_spatial_mean_fwd(L.dev(x, "input"), L.dev(out), N * C, H * W, scale, L.stream()),
                "dvf_spatial_mean_fwd")
        ctx.shape, ctx.scale = (N, C, H, W), scale
        return out

    @staticmethod
    def backward(ctx, gout):
        N, C, H, W = ctx.shape
        gin = torch.empty((N, C, H, W), device=gout.device, dtype=torch.float32)
        L.check(L.lib().dvf_spatial_mean_bwd(L.dev(_c(gout), "grad_out"), L.dev(gin), N * C, H * W, ctx.scale,
                                             L.stream()), "dvf_spatial_mean_bwd")
        return gin, None


class DepthwiseUp2xFn(torch.autograd.Function):
    """skip + ConvTranspose2d(C, C, 4, stride=2, padding=1, groups=C)(x): one top-down step of FeatExtractor
    (reference feat_extractor.py:72-82)."""

    @staticmethod
    def forward(ctx, x, weight, bias, skip):
        ctx.wparam = weight if getattr(weight, "_dvf_grad", None) is not None else None
        ctx.bparam = bias if getattr(bias, "_dvf_grad", None) is not None else None
        x, weight, bias, skip = _c(x), _c(weight), _c(bias), _c(skip)
        N, C, H, W = x.shape
        if tuple(skip.shape) != (N, C, 2 * H, 2 * W) or tuple(weight.shape) != (C, 1, 4, 4):
            raise ValueError(f"shape mismatch: x {tuple(x.shape)} weight {tuple(weight.shape)} skip {tuple(skip.shape)}")
        out = torch.empty_like(skip)
        L.check(L.lib().dvf_dwconvt4x4s2_fwd(L.dev(x, "input"), L.dev(weight, "weight"), L.dev(bias, "bias"),
                                             L.dev(skip, "skip"), L.dev(out), N, C, H, W, L.stream()), "dvf_dwconvt4x4s2_fwd")
        ctx.save_for_backward(x, weight)
        return out

    @staticmethod
    def backward(ctx, gout):
        x, weight = ctx.saved_tensors
        N, C, H, W = x.shape
        gout = _c(gout)
        dx = torch.empty_like(x) if ctx.needs_input_grad[0] else None
        dw = torch.empty_like(weight) if ctx.needs_input_grad[1] else None
        db = torch.empty(C, device=x.device) if (dw is not None and ctx.needs_input_grad[2]) else None
        L.check(L.lib().dvf_dwconvt4x4s2_bwd(L.dev(x), L.dev(weight), L.dev(gout, "grad_out"), L.dev(dx), L.dev(dw), L.dev(db),
                                             N, C, H, W, L.stream()), "dvf_dwconvt4x4s2_bwd")
        if dw is not None and ctx.wparam is not None:
            ctx.wparam._dvf_grad.add_(dw)
            ctx.wparam._dvf_owner.grad_ready(ctx.wparam)
            dw = None
        if db is not None and ctx.bparam is not None:
            ctx.bparam._dvf_grad.add_(db)
            ctx.bparam._dvf_owner.grad_ready(ctx.bparam)
            db = None
        return dx, dw, db, (gout if ctx.needs_input_grad[3] else None)


def reciprocal(x, eps=0.0):
    return RecipFn.apply(x, float(eps))


def area_downsample(x, out_hw):
    """F.interpolate(x, out_hw, mode='area') for inputs that need no gradient (images)."""
    x = _c(x)
    N, C, H, W = x.shape
    if (H, W) == tuple(out_hw):
        return x
    out = torch.empty((N, C, out_hw[0], out_hw[1]), device=x.device, dtype=torch.float32)
    L.check(L.lib().dvf_area_downsample(L.dev(x, "input"), L.dev(out), N * C, H, W, out_hw[0], out_hw[1], L.stream()),
            "dvf_area_downsample")
    return out


def bilinear_half(x):
    """F.interpolate(x, scale_factor=0.5, mode='bilinear') for inputs that need no gradient (images)."""
    x = _c(x)
    N, C, H, W = x.shape
    oh, ow = H // 2, W // 2
    out = torch.empty((N, C, oh, ow), device=x.device, dtype=torch.float32)
    L.check(L.lib().dvf_resize_bilinear_fwd(L.dev(x, "input"), L.dev(out), N * C, H, W, oh, ow, 2.0, 2.0, L.stream()),
            "dvf_resize_bilinear_fwd")
    return out


# ------------------------------------------------------------------------------------------ module shells

class FusedConv2d(nn.Module):
    """nn.Conv2d (+ fused activation) with torch's parameter layout [C_out, C_in, k, k]."""
    transposed = False

    def __init__(self, in_planes, out_planes, kernel_size, stride=1, padding=0, act=L.ACT_NONE, alpha=1.0, beta=0.0,
                 output_padding=0, fuse_bwd=False):
        """fuse_bwd (ReLU layers only): the network GUARANTEES that every consumer of this layer's output is a
        FusedConv2d / FusedConvTranspose2d taking the returned tensor itself (no view, no other op) -- see ReluTag."""
        super().__init__()
        self.fuse_bwd = bool(fuse_bwd) and act == L.ACT_RELU
        self.k, self.stride, self.pad, self.opad = kernel_size, stride, padding, output_padding
        self.act, self.alpha, self.beta = act, float(alpha), float(beta)
        self.in_planes, self.out_planes = in_planes, out_planes
        shape = (in_planes, out_planes) if self.transposed else (out_planes, in_planes)
        self.weight = nn.Parameter(torch.empty(*shape, kernel_size, kernel_size))
        self.bias = nn.Parameter(torch.empty(out_planes))
        self.reset_parameters()

    def reset_parameters(self):
        # torch's default Conv2d init (kaiming_uniform(a=sqrt(5)) == U(-1/sqrt(fan_in), 1/sqrt(fan_in)))
        fan_in = self.weight.shape[1] * self.k * self.k
        bound = 1.0 / math.sqrt(fan_in)
        with torch.no_grad():
            self.weight.uniform_(-bound, bound)
            self.bias.uniform_(-bound, bound)

    def forward(self, *inputs, out_hw=None):
        tag = None
        if self.fuse_bwd and FUSE_RELU_BWD and torch.is_grad_enabled() and self.bias.requires_grad and inputs[0].is_cuda:
            tag = ReluTag()
        cfg = (self.k, self.stride, self.pad, self.opad, self.transposed, self.act, self.alpha, self.beta, out_hw, tag)
        out = ConvFn.apply(self.weight, self.bias, cfg, *inputs)
        if tag is not None:
            out._dvf_relu_tag = tag
        return out

    def extra_repr(self):
        return (f"{self.in_planes}, {self.out_planes}, kernel_size={self.k}, stride={self.stride}, padding={self.pad}, "
                f"act={self.act}")


class FusedConvTranspose2d(FusedConv2d):
    """nn.ConvTranspose2d (+ fused activation), parameter layout [C_in, C_out, k, k]."""
    transposed = True


class DepthwiseUp2x(nn.Module):
    """nn.ConvTranspose2d(C, C, kernel_size=4, padding=1, stride=2, groups=C) with the following residual add fused;
    parameter layout [C, 1, 4, 4] as torch's."""

    def __init__(self, channels):
        super().__init__()
        self.channels = channels
        self.weight = nn.Parameter(torch.empty(channels, 1, 4, 4).uniform_(-0.25, 0.25))
        self.bias = nn.Parameter(torch.empty(channels).uniform_(-0.25, 0.25))

    def forward(self, x, skip):
        return DepthwiseUp2xFn.apply(x, self.weight, self.bias, skip)


class FusedAct(nn.Module):
    """Index placeholder: keeps the reference's nn.Sequential numbering (conv1.0 / conv1.2 ...) while the
    activation itself runs in the epilogue of the preceding convolution kernel."""

    def forward(self, x):
        return x


def xavier_init_(module):
    """init_weights() of the reference nets: xavier_uniform_ on every conv weight, zero bias
    (DispNetS.py:81-86, PoseExpNet_sfm.py:51-56, feat_extractor.py:85-90)."""
    for m in module.modules():
        if isinstance(m, (FusedConv2d, DepthwiseUp2x)):
            nn.init.xavier_uniform_(m.weight)
            nn.init.zeros_(m.bias)
