"""CPU-side checks of the drop-in boundary: the C-ABI library loads and exports every symbol that
include/dvf_hip.h declares, and the host binding's signature table covers exactly those symbols.
No compute call is made (no GPU here)."""
import os
import re

import pytest

from conftest import ROOT
from dvf import lib as L


def _declared():
    text = open(os.path.join(ROOT, "include", "dvf_hip.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(dvf_[a-z0-9_]+)\s*\(", text)))


def test_library_exports_every_declared_symbol():
    names = _declared()
    assert len(names) >= 10
    lib = L.lib()
    for n in names:
        assert hasattr(lib, n), f"libdvf_hip.so does not export {n}"
    assert sorted(L.SIGNATURES) == names, "host binding table and header disagree"


def test_version_and_error_strings():
    lib = L.lib()
    assert lib.dvf_version() >= 100
    assert b"invalid" in lib.dvf_error_string(-1)
    assert lib.dvf_error_string(0) == b"ok"


def test_null_arguments_are_rejected_without_touching_the_gpu():
    lib = L.lib()
    assert lib.dvf_inverse_warp_fwd(None, None, None, None, None, None, 1, 3, 8, 8, 0, None) == -1
    assert lib.dvf_smooth_loss_fwd(None, None, None, 1, 8, 8, 1.0, 0, None) == -1


def test_product_refuses_cpu_tensors():
    """The product path has no CPU fallback: a CPU tensor must raise, not silently run elsewhere."""
    import torch
    import loss_functions
    d = torch.rand(1, 1, 8, 8)
    with pytest.raises(RuntimeError, match="no CPU fallback"):
        loss_functions.smooth_loss(d)


def _desc(n, cin, h, w, cout, k=3, stride=1, pad=1, transposed=0):
    oh = (h - 1) * stride - 2 * pad + k + (1 if transposed and stride == 2 and k == 3 else 0) if transposed else (h + 2 * pad - k) // stride + 1
    ow = (w - 1) * stride - 2 * pad + k + (1 if transposed and stride == 2 and k == 3 else 0) if transposed else (w + 2 * pad - k) // stride + 1
    return L.ConvDesc(n, cin, h, w, cout, oh, ow, k, k, stride, pad, transposed, L.ACT_RELU, 1.0, 0.0)


def test_packed_weight_planner_is_host_only_and_consistent():
    """The planning half of the packed-weight protocol (sizes, job records) runs on the host: check it here."""
    import ctypes
    lib = L.lib()
    d = _desc(4, 128, 32, 104, 128)
    segs = L.int_array([128])
    nf = lib.dvf_conv2d_packed_floats(ctypes.byref(d), segs, 1, 0)
    assert nf >= 128 * 128 * 9 and nf % 256 == 0                 # at least the weights, whole 1 KiB pieces
    assert lib.dvf_conv2d_packed_floats(ctypes.byref(d), segs, 1, 1) >= 128 * 128 * 9
    assert lib.dvf_conv2d_ws_floats(ctypes.byref(d), segs, 1, 0) >= 0
    assert lib.dvf_conv2d_packed_floats(ctypes.byref(d), segs, 1, 7) == -1          # bad op_kind
    # at most 16 output channels: stays on the unpacked kernels (17..32 run the pipelined kernel since round 3)
    small = _desc(4, 17, 256, 832, 16)
    assert lib.dvf_conv2d_packed_floats(ctypes.byref(small), L.int_array([16, 1]), 2, 0) == L.ERR_UNSUPPORTED
    mid = _desc(4, 65, 128, 416, 32)
    assert lib.dvf_conv2d_packed_floats(ctypes.byref(mid), L.int_array([32, 32, 1]), 3, 0) >= 32 * 65 * 9
    # deterministic weight gradient: scratch size is a host-side plan too
    assert lib.dvf_conv2d_wgrad_ws_floats(ctypes.byref(d), segs, 1) > 0
    # virtual concat with a 1-channel segment: the wide segments are packed for dgrad, the narrow one is not
    cat = _desc(4, 257, 32, 104, 128)
    segs3 = L.int_array([128, 128, 1])
    n_dgrad = lib.dvf_conv2d_packed_floats(ctypes.byref(cat), segs3, 3, 1)
    assert n_dgrad >= 2 * 128 * 128 * 9
    # job records: one per packed op instance, block counts positive (pointers are recorded, not dereferenced)
    buf = (ctypes.c_char * (L.PACK_JOB_BYTES * 3))()
    blocks = (ctypes.c_int * 3)()
    lds = ctypes.c_int(4)
    fake = ctypes.c_void_p(0x1000)
    n = lib.dvf_conv2d_pack_jobs(ctypes.byref(cat), segs3, 3, 1, fake, fake, ctypes.cast(buf, ctypes.c_void_p), 3, blocks,
                                 ctypes.byref(lds))
    assert n == 2 and blocks[0] > 0 and blocks[1] > 0 and 4 < lds.value <= 64 * 1024
    # strided transposed convolution (4 output-parity classes share one packed buffer)
    up = _desc(4, 256, 16, 52, 128, 3, 2, 1, 1)
    assert lib.dvf_conv2d_packed_floats(ctypes.byref(up), L.int_array([256]), 1, 0) >= 256 * 128 * 9


def test_bench_spawn_parent_makes_no_gpu_runtime_call():
    """`python bench.py --gpus N` without a launcher fork+execs its N ranks: the parent must not have loaded torch (whose
    device count goes through HIP/HSA) -- on the GPU pool an exec from a GPU-initialised process takes the machine down.
    It must also notice a failed rank, stop the others and exit non-zero.  (Here: no GPU, so every rank exits at once.)"""
    import subprocess
    import sys
    code = (
        "import atexit, sys, runpy\n"
        "atexit.register(lambda: sys.stderr.write('PARENT_TORCH_IMPORTED\\n' if 'torch' in sys.modules else 'PARENT_TORCH_FREE\\n'))\n"
        f"sys.argv = [{os.path.join(ROOT, 'bench.py')!r}, '--gpus', '2', '--steps', '1', '--warmup', '0']\n"
        f"runpy.run_path({os.path.join(ROOT, 'bench.py')!r}, run_name='__main__')\n")
    r = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=300)
    assert "PARENT_TORCH_FREE" in r.stderr, r.stderr[-2000:]
    assert r.returncode != 0            # no GPU in the CPU container: the ranks refuse to run, the parent reports it
    assert r.stdout.strip() == ""       # and prints no JSON line
