"""GPU side of the input pipeline: ``gpu_imresize`` reproduces, on the device and bit for bit, what
the reference's loaders do to every frame on the host -- ``scipy.misc.imresize(img.astype(float32), (H, W))`` =
bytescale + PIL BILINEAR resize (un_dataset.py:63-66, dataset.py:50-51) -- starting from the raw uint8 frame, so a
loader only has to decode and upload.  The coefficient tables follow Pillow's ``precompute_coeffs`` and
``normalize_coeffs_8bpc`` (src/libImaging/Resample.c) and are cached per size pair."""
import ctypes
import functools
import math

import numpy as np
import torch

from . import lib as L

_PRECISION_BITS = 32 - 8 - 2


@functools.lru_cache(maxsize=64)
def _coeffs(in_size, out_size):
    """(bounds [out,2] int32, coefficients [out,ksize] int32, ksize) of PIL's bilinear (triangle) filter."""
    scale = in_size / out_size
    filterscale = max(scale, 1.0)
    support = 1.0 * filterscale
    ksize = int(math.ceil(support)) * 2 + 1
    bounds = np.zeros((out_size, 2), dtype=np.int32)
    kk = np.zeros((out_size, ksize), dtype=np.float64)
    ss = 1.0 / filterscale
    for xx in range(out_size):
        center = (xx + 0.5) * scale
        xmin = max(int(center - support + 0.5), 0)
        xmax = min(int(center + support + 0.5), in_size) - xmin
        x = np.arange(xmax, dtype=np.float64)
        w = np.maximum(0.0, 1.0 - np.abs((x + xmin - center + 0.5) * ss))
        ww = w.sum()
        if ww != 0.0:
            w = w / ww
        kk[xx, :xmax] = w
        bounds[xx] = (xmin, xmax)
    fixed = np.where(kk < 0, np.trunc(-0.5 + kk * (1 << _PRECISION_BITS)), np.trunc(0.5 + kk * (1 << _PRECISION_BITS))).astype(np.int32)
    return bounds, fixed, ksize


_TABLES = {}


def _device_tables(in_size, out_size, device):
    key = (in_size, out_size, str(device))
    t = _TABLES.get(key)
    if t is None:
        b, k, ks = _coeffs(in_size, out_size)
        _TABLES[key] = t = (torch.from_numpy(b).to(device), torch.from_numpy(k).to(device), ks)
    return t


def gpu_imresize(img_u8_hwc, size):
    """uint8 [H0, W0, C] device tensor (a decoded frame) -> float32 [C, H, W] with the values of
    ``imresize(frame.astype(float32), size)`` (integers 0..255), i.e. one tensor of the reference dataset tuple."""
    if not img_u8_hwc.is_cuda or img_u8_hwc.dtype != torch.uint8 or img_u8_hwc.dim() != 3 or not img_u8_hwc.is_contiguous():
        raise ValueError("gpu_imresize needs a contiguous uint8 [H, W, C] tensor on the GPU; there is no CPU fallback")
    IH, IW, C = img_u8_hwc.shape
    OH, OW = int(size[0]), int(size[1])
    dev = img_u8_hwc.device
    hb, hk, hks = _device_tables(IW, OW, dev)
    vb, vk, vks = _device_tables(IH, OH, dev)
    tmp = torch.empty((IH, OW, C), dtype=torch.uint8, device=dev)
    mm = torch.empty(2, dtype=torch.int32, device=dev)
    out = torch.empty((C, OH, OW), dtype=torch.float32, device=dev)
    p = lambda t: ctypes.c_void_p(t.data_ptr())
    L.check(L.lib().dvf_imresize_u8(p(img_u8_hwc), IH, IW, C, p(hb), p(hk), hks, p(vb), p(vk), vks, p(tmp), p(mm), p(out), OH, OW,
                                    L.stream()), "dvf_imresize_u8")
    return out
