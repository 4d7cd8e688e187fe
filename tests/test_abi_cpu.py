"""CPU-side checks of the drop-in boundary: the library builds, loads, exports
every symbol include/nubovca.h declares, and refuses to work without a GPU
(no CPU fallback in the product path)."""
import ctypes as C
import os
import re

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def lib():
    import __graft_entry__ as ge
    ge.build()
    from nubovca import capi
    return capi.load()


def test_header_symbols_all_exported(lib):
    from nubovca import capi
    hdr = open(os.path.join(ROOT, "include", "nubovca.h")).read()
    declared = set(re.findall(r"\b(nvca_[a-z0-9_]+)\s*\(", hdr))
    assert declared == set(capi.SYMBOLS), declared ^ set(capi.SYMBOLS)
    for s in sorted(declared):
        assert hasattr(lib, s), s


def test_no_cpu_fallback(lib):
    import torch
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    h = C.c_void_p()
    rc = lib.nvca_ctx_create(0, C.byref(h))
    assert rc == -2 and not h.value          # NVCA_ERR_NO_DEVICE


def test_product_does_not_import_oracle():
    pkg = os.path.join(ROOT, "nubomedia-vca_amd")
    for dp, _, fs in os.walk(pkg):
        for f in fs:
            if f.endswith((".py", ".cpp", ".hip", ".h", ".c")):
                txt = open(os.path.join(dp, f), errors="ignore").read()
                assert "oracle" not in txt.lower() or f == "synth.py" and False, os.path.join(dp, f)
