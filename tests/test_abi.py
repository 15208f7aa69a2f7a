"""The C-ABI library loads and exports every symbol include/*.h declares (no compute without a GPU)."""
import ctypes
import os
import re

import pytest

from gatekeeper_amd import _lib as L

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def header_functions():
    """every function any include/*.h declares"""
    names = set()
    for h in sorted(os.listdir(os.path.join(ROOT, "include"))):
        if not h.endswith(".h"):
            continue
        src = open(os.path.join(ROOT, "include", h)).read()
        src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
        names |= set(re.findall(r"\b(gk_[a-z0-9_]+)\s*\(", src))
    return sorted(names)


def test_header_matches_binding():
    assert header_functions() == sorted(L.EXPORTS)


def test_product_library_exports_every_symbol():
    path = L.library_path(False)
    assert os.path.exists(path), "build it: python -c 'import __graft_entry__ as g; g.build()'"
    lib = ctypes.CDLL(path)
    for name in header_functions():
        assert hasattr(lib, name), name
    assert b"gfx950" in ctypes.cast(lib.gk_version, ctypes.CFUNCTYPE(ctypes.c_char_p))()


def test_product_fails_loudly_without_gpu():
    """No CPU fallback: without a HIP device gk_engine_create must fail with GK_ERR_DEVICE."""
    import torch
    if torch.cuda.is_available():
        pytest.skip("a GPU is visible")
    from gatekeeper_amd.driver import Engine, EngineError
    with pytest.raises(EngineError) as ei:
        Engine(hostemu=False)
    assert ei.value.code == L.GK_ERR_DEVICE


def test_product_library_has_no_emulator():
    """The CPU emulation lives only in tests/native; the product .so must not contain it."""
    import subprocess
    out = subprocess.run(["nm", "-D", "--defined-only", L.library_path(False)], capture_output=True, text=True).stdout
    assert "VecAcc" not in out
