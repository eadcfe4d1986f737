"""ctypes binding of libdvf_hip.so (C ABI declared in include/dvf_hip.h).

The library is the product: there is no CPU or stock-torch fallback.  If the shared object is
missing or a call returns an error code, this module raises.
"""
import ctypes
import os

import torch

_HERE = os.path.dirname(os.path.abspath(__file__))
# (DVF_LIB: tools/ load the -DDVF_TUNING build, libdvf_hip_tuning.so, through this; the product path never sets it)
LIB_PATH = os.environ.get("DVF_LIB") or os.path.join(_HERE, "libdvf_hip.so")
_lib = None

c_fp = ctypes.c_void_p          # device pointer to float
c_i = ctypes.c_int
c_u32 = ctypes.c_uint32
c_f = ctypes.c_float
c_i64 = ctypes.c_int64
c_pp = ctypes.POINTER(ctypes.c_void_p)   # host array of device pointers



class ConvDesc(ctypes.Structure):
    """struct dvf_conv_desc of include/dvf_hip.h"""
    _fields_ = [(n, ctypes.c_int) for n in ("N", "C_in", "H_in", "W_in", "C_out", "H_out", "W_out", "KH", "KW",
                                             "stride", "pad", "transposed", "act")] + \
               [("alpha", ctypes.c_float), ("beta", ctypes.c_float)]


c_desc = ctypes.POINTER(ConvDesc)
c_ip = ctypes.POINTER(ctypes.c_int)

ROT_QUAT, PAD_BORDER, ALIGN_CORNERS, POSE_SE3, PIXEL_COORDS, CAFFE_ABSLOSS = 1, 2, 4, 8, 16, 32
ACT_NONE, ACT_RELU, ACT_SIGMOID_AFFINE = 0, 1, 2
MAX_VIEWS, MAX_SEGS = 4, 5

# name -> (restype, argtypes); kept in one table so tests can check every declared symbol is exported
SIGNATURES = {
    "dvf_version": (c_i, []),
    "dvf_error_string": (ctypes.c_char_p, [c_i]),
    "dvf_build_has_tuning": (c_i, []),
    "dvf_conv2d_last_plans": (c_i, [c_ip, c_i]),
    "dvf_inverse_warp_fwd": (c_i, [c_fp] * 6 + [c_i] * 4 + [c_u32, c_fp]),
    "dvf_inverse_warp_bwd": (c_i, [c_fp] * 10 + [c_i] * 4 + [c_u32, c_fp]),
    "dvf_pose_ws_floats": (c_i64, [c_i, c_i]),
    "dvf_pose_vec2mat_fwd": (c_i, [c_fp, c_fp, c_i, c_u32, c_fp]),
    "dvf_pose_vec2mat_bwd": (c_i, [c_fp] * 4 + [c_i, c_u32, c_fp]),
    "dvf_pixel2cam_fwd": (c_i, [c_fp] * 3 + [c_i] * 3 + [c_fp]),
    "dvf_pixel2cam_bwd": (c_i, [c_fp] * 3 + [c_i] * 3 + [c_fp]),
    "dvf_cam2pixel_fwd": (c_i, [c_fp] * 4 + [c_i] * 3 + [c_u32, c_fp]),
    "dvf_cam2pixel_bwd": (c_i, [c_fp] * 6 + [c_i] * 3 + [c_u32, c_fp]),
    "dvf_photo_loss_fwd": (c_i, [c_fp, c_pp, c_i] + [c_fp] * 8 + [c_i] * 4 + [c_f, c_u32, c_fp]),
    "dvf_photo_partials_floats": (c_i64, [c_i] * 4),
    "dvf_photo_pose_ws_floats": (c_i64, [c_i] * 4),
    "dvf_photo_loss_bwd": (c_i, [c_fp, c_pp, c_i] + [c_fp] * 9 + [c_pp, c_fp, c_fp] + [c_i] * 4 + [c_f, c_u32, c_fp]),
    "dvf_imresize_u8": (c_i, [c_fp, c_i, c_i, c_i, c_fp, c_fp, c_i, c_fp, c_fp, c_i, c_fp, c_fp, c_fp, c_i, c_i, c_fp]),
    "dvf_edge_smooth_fwd": (c_i, [c_fp] * 4 + [c_i] * 4 + [c_f, c_f, c_f, c_i, c_fp]),
    "dvf_edge_smooth_partials_floats": (c_i64, [c_i] * 3),
    "dvf_edge_smooth_bwd": (c_i, [c_fp] * 4 + [c_i] * 4 + [c_f, c_f, c_f, c_fp]),
    "dvf_smooth_loss_fwd": (c_i, [c_fp] * 3 + [c_i] * 3 + [c_f, c_i, c_fp]),
    "dvf_smooth_partials_floats": (c_i64, [c_i] * 3),
    "dvf_smooth_loss_bwd": (c_i, [c_fp] * 3 + [c_i] * 3 + [c_f, c_fp]),
    "dvf_conv2d_fwd": (c_i, [c_desc, c_pp, c_ip, c_i, c_fp, c_fp, c_fp, c_fp]),
    "dvf_conv2d_dgrad": (c_i, [c_desc, c_fp, c_fp, c_pp, c_ip, c_i, c_fp]),
    "dvf_conv2d_fwd_ws": (c_i, [c_desc, c_pp, c_ip, c_i, c_fp, c_fp, c_fp, c_fp, c_i64, c_fp]),
    "dvf_conv2d_dgrad_ws": (c_i, [c_desc, c_fp, c_fp, c_pp, c_ip, c_i, c_fp, c_i64, c_fp]),
    "dvf_conv2d_wgrad": (c_i, [c_desc, c_pp, c_ip, c_i, c_fp, c_fp, c_i, c_fp]),
    "dvf_conv2d_wgrad_bias": (c_i, [c_desc, c_pp, c_ip, c_i, c_fp, c_fp, c_i, c_fp, c_i, c_fp]),
    "dvf_conv2d_wgrad_ws_floats": (c_i64, [c_desc, c_ip, c_i]),
    "dvf_conv2d_wgrad_det": (c_i, [c_desc, c_pp, c_ip, c_i, c_fp, c_fp, c_i, c_fp, c_i, c_fp, c_i64, c_fp]),
    "dvf_conv2d_packed_floats": (c_i64, [c_desc, c_ip, c_i, c_i]),
    "dvf_conv2d_pack": (c_i, [c_desc, c_ip, c_i, c_i, c_fp, c_fp, c_fp]),
    "dvf_conv2d_ws_floats": (c_i64, [c_desc, c_ip, c_i, c_i]),
    "dvf_conv2d_pack_jobs": (c_i, [c_desc, c_ip, c_i, c_i, c_fp, c_fp, c_fp, c_i, c_ip, c_ip]),
    "dvf_conv2d_pack_batch": (c_i, [c_fp, c_fp, c_fp, c_i, c_i, c_i, c_fp]),
    "dvf_conv2d_fwd_packed": (c_i, [c_desc, c_pp, c_ip, c_i, c_fp, c_fp, c_fp, c_fp, c_i64, c_fp]),
    "dvf_conv2d_dgrad_packed": (c_i, [c_desc, c_fp, c_fp, c_fp, c_pp, c_ip, c_i, c_fp, c_i64, c_fp]),
    "dvf_conv2d_dgrad_masked": (c_i, [c_desc, c_fp, c_fp, c_fp, c_pp, c_ip, c_i, c_fp, c_i64, c_pp, c_pp, c_fp]),
    "dvf_act_bwd": (c_i, [c_fp] * 4 + [c_i] * 4 + [c_f, c_f, c_fp]),
    "dvf_act_bwd2": (c_i, [c_fp] * 4 + [c_i] * 4 + [c_f, c_f, c_i, c_fp]),
    "dvf_act_bwd_ws_floats": (c_i64, [c_i, c_i, c_i]),
    "dvf_act_bwd_det": (c_i, [c_fp] * 4 + [c_i] * 4 + [c_f, c_f, c_i, c_fp, c_i64, c_fp]),
    "dvf_resize_bilinear_fwd": (c_i, [c_fp, c_fp] + [c_i] * 5 + [c_f, c_f, c_fp]),
    "dvf_upsample2x_bwd": (c_i, [c_fp, c_fp] + [c_i] * 5 + [c_fp]),
    "dvf_recip_fwd": (c_i, [c_fp, c_fp, c_f, c_i64, c_fp]),
    "dvf_recip_bwd": (c_i, [c_fp, c_fp, c_fp, c_i64, c_fp]),
    "dvf_spatial_mean_fwd": (c_i, [c_fp, c_fp, c_i, c_i, c_f, c_fp]),
    "dvf_spatial_mean_bwd": (c_i, [c_fp, c_fp, c_i, c_i, c_f, c_fp]),
    "dvf_area_downsample": (c_i, [c_fp, c_fp] + [c_i] * 5 + [c_fp]),
    "dvf_dwconvt4x4s2_fwd": (c_i, [c_fp] * 5 + [c_i] * 4 + [c_fp]),
    "dvf_dwconvt4x4s2_bwd": (c_i, [c_fp] * 6 + [c_i] * 4 + [c_fp]),
    "dvf_bce_ones_fwd": (c_i, [c_fp, c_fp, c_fp, c_i64, c_i, c_fp]),
    "dvf_bce_ones_bwd": (c_i, [c_fp, c_fp, c_fp, c_i64, c_fp]),
    "dvf_adam_step": (c_i, [c_fp] * 4 + [c_i64, c_fp, c_i] + [c_f] * 5 + [c_fp]),
}


def lib():
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise RuntimeError(
                f"{LIB_PATH} not found: build it with `make -C depth-vo-feat_amd/csrc` "
                "(or __graft_entry__.build()).  There is no fallback path.")
        L = ctypes.CDLL(LIB_PATH)
        for name, (res, args) in SIGNATURES.items():
            fn = getattr(L, name)           # AttributeError if the .so does not export a declared symbol
            fn.restype, fn.argtypes = res, args
        _lib = L
    return _lib


def check(rc, what):
    if rc != 0:
        raise RuntimeError(f"{what} failed: {lib().dvf_error_string(rc).decode()} (code {rc})")


_DEV_INDEX = None


def stream():
    """Raw handle of torch's current stream on this process's device (one process per GPU): the C ABI enqueues on it.
    (torch.cuda.current_stream() costs ~10 us of host time per call; this path ~0.3 us.)"""
    global _DEV_INDEX
    if _DEV_INDEX is None:
        _DEV_INDEX = torch.cuda.current_device()
    return ctypes.c_void_p(torch._C._cuda_getCurrentRawStream(_DEV_INDEX))


def stream_raw():
    """stream() as a plain integer (ctypes converts it for a void* parameter)."""
    global _DEV_INDEX
    if _DEV_INDEX is None:
        _DEV_INDEX = torch.cuda.current_device()
    return torch._C._cuda_getCurrentRawStream(_DEV_INDEX)


def dev(t, name="tensor"):
    """Validate a tensor for the C ABI (cuda, fp32, contiguous) and return its device pointer (an integer; None for None)."""
    if t is None:
        return None
    if not t.is_cuda:
        raise RuntimeError(f"{name}: the HIP path needs a CUDA/HIP tensor, got {t.device}; there is no CPU fallback")
    if t.dtype != torch.float32:
        raise TypeError(f"{name}: expected float32, got {t.dtype}")
    if not t.is_contiguous():
        raise RuntimeError(f"{name}: expected a contiguous tensor")
    return t.data_ptr()


def ptr_array(tensors, name="tensors"):
    arr = (ctypes.c_void_p * len(tensors))()
    for i, t in enumerate(tensors):
        if t is not None:
            arr[i] = dev(t, name)
    return arr


def int_array(vals):
    return (ctypes.c_int * len(vals))(*vals)


def geom_flags(rotation_mode="euler", padding_mode="zeros", align_corners=False):
    if rotation_mode not in ("euler", "quat"):
        raise ValueError(f"rotation_mode must be 'euler' or 'quat', got {rotation_mode!r}")
    if padding_mode not in ("zeros", "border"):
        raise ValueError(f"padding_mode must be 'zeros' or 'border', got {padding_mode!r}")
    return ((ROT_QUAT if rotation_mode == "quat" else 0) | (PAD_BORDER if padding_mode == "border" else 0) |
            (ALIGN_CORNERS if align_corners else 0))


class KernelTimer:
    """Optional per-call timing with HIP events on the stream the kernels are launched on (torch's current
    stream).  ``bench.py`` switches it on for one eager pass to measure per-kernel durations next to their
    algorithmic FLOPs / bytes; it is off (None) otherwise and costs nothing."""

    FAMILY = {1: "pipe", 2: "gather", 3: "head", 4: "head", 5: "head", 6: "wgrad", 7: "head", 8: "wgrad_pipe", 9: "dconvt"}

    def __init__(self):
        self.records = []       # (kind, start_event, stop_event, flops, bytes, tag, kernel family)
        self._kernels = []      # per record: the kernel families the call launched, in launch order

    def records_with_kernels(self):
        torch.cuda.synchronize()
        return [r + (list(k),) for r, k in zip(self.records, self._kernels)]

    def start(self):
        ev = torch.cuda.Event(enable_timing=True)
        ev.record(torch.cuda.current_stream())
        return ev

    def stop(self, kind, start_ev, flops=0.0, nbytes=0.0, tag=""):
        ev = torch.cuda.Event(enable_timing=True)
        ev.record(torch.cuda.current_stream())
        fam, kernels = "", ()
        if kind.startswith("conv_"):
            # which kernels the call launched (dvf_conv2d_last_plans): the kernel family with the most work names the call
            buf = (ctypes.c_int * 96)()
            n = lib().dvf_conv2d_last_plans(buf, 96)
            fams = [self.FAMILY.get(buf[i], "other") for i in range(0, n, 12)]
            fam = next((f for f in ("pipe", "wgrad_pipe", "gather", "wgrad", "dconvt", "head") if f in fams), "other")
            kernels = tuple(fams)
        self.records.append((kind, start_ev, ev, float(flops), float(nbytes), tag, fam))
        self._kernels.append(kernels)

    def table(self):
        """Per-call rows (kind, tag, ms, TFLOP/s, GB/s), slowest first."""
        torch.cuda.synchronize()
        rows = []
        for kind, a, b, fl, by, tag, fam in self.records:
            ms = a.elapsed_time(b)
            rows.append((ms, kind, (fam + " " if fam else "") + tag, fl / (ms * 1e-3) / 1e12 if ms > 0 else 0.0,
                         by / (ms * 1e-3) / 1e9 if ms > 0 else 0.0))
        return sorted(rows, reverse=True)

    def summary(self):
        """Totals per kind, and per "kind/family" for the convolution calls (family: the kernel that ran)."""
        torch.cuda.synchronize()
        out = {}
        for kind, a, b, fl, by, _tag, fam in self.records:
            ms = a.elapsed_time(b)
            for key in ((kind, kind + "/" + fam) if fam else (kind,)):
                d = out.setdefault(key, {"calls": 0, "ms": 0.0, "flops": 0.0, "bytes": 0.0})
                d["calls"] += 1
                d["ms"] += ms
                d["flops"] += fl
                d["bytes"] += by
        return out


PLAN_LOG = None  # set to a set(): every convolution call adds (op, plan record) of the kernels it launched


def note_plans(op):
    """Read the plan records of the convolution call just made on this thread into PLAN_LOG (tests only)."""
    if PLAN_LOG is None:
        return
    buf = (ctypes.c_int * 96)()
    n = lib().dvf_conv2d_last_plans(buf, 96)
    for i in range(0, n, 12):
        PLAN_LOG.add((op,) + tuple(buf[i:i + 12]))


TIMER = None    # set to a KernelTimer() to record
ERR_UNSUPPORTED = -3
PACK_JOB_BYTES = 512
SERIALIZE = os.environ.get("DVF_SERIALIZE", "0") == "1"   # True: no side streams (per-kernel timing passes, debugging)
AUX_STREAMS = {}      # device (pose network, dvf/steps.py) or (device, name) -> auxiliary compute stream


def aux_stream(device, name="pose"):
    """Named auxiliary compute stream of a device: "pose" (the pose network runs beside the depth network, dvf/steps.py),
    "heads" (the disparity heads run beside the next up-convolution, DispNetS.py)."""
    key = device if name == "pose" else (device, name)
    s = AUX_STREAMS.get(key)
    if s is None:
        AUX_STREAMS[key] = s = torch.cuda.Stream(device=device)
    return s


def aux_streams_on(device):
    return [s for k, s in AUX_STREAMS.items() if k == device or (isinstance(k, tuple) and k[0] == device)]


def join_aux_streams():
    """Make the current stream wait for everything enqueued on the auxiliary compute streams."""
    if not AUX_STREAMS:
        return
    cur = torch.cuda.current_stream()
    for s in aux_streams_on(cur.device):
        cur.wait_stream(s)


PACK_EPOCH = 0  # manual invalidation of every packed weight copy (FlatAdam bumps a per-parameter epoch instead: dvf/conv.py::_stamp)


def timed(kind, flops=0.0, nbytes=0.0, tag=""):
    """Context manager: times the enclosed launches when TIMER is active."""
    return _Timed(kind, flops, nbytes, tag)


class _Timed:
    __slots__ = ("kind", "flops", "nbytes", "ev", "tag")

    def __init__(self, kind, flops, nbytes, tag):
        self.kind, self.flops, self.nbytes, self.ev, self.tag = kind, flops, nbytes, None, tag

    def __enter__(self):
        if TIMER is not None:
            self.ev = TIMER.start()
        return self

    def __exit__(self, *exc):
        if self.ev is not None and TIMER is not None:
            TIMER.stop(self.kind, self.ev, self.flops, self.nbytes, self.tag)
        return False
