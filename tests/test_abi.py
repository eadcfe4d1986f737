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
