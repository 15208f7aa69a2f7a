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


def _build_shim(tmp_path, lib_path):
    """tests/native/abi_shim.c (plain C99: what cgo would bind) against `lib_path`, warnings as errors"""
    import subprocess
    exe = os.path.join(str(tmp_path), "abi_shim")
    d, name = os.path.dirname(lib_path), os.path.basename(lib_path)
    cmd = ["gcc", "-std=c99", "-Wall", "-Wextra", "-pedantic", "-Werror", "-I" + os.path.join(ROOT, "include"), os.path.join(ROOT, "tests", "native", "abi_shim.c"),
           "-o", exe, "-L" + d, "-l:" + name, "-lpthread", "-Wl,-rpath," + d]
    r = subprocess.run(cmd, capture_output=True, text=True)
    assert r.returncode == 0, r.stderr[-3000:]
    return exe


def test_headers_are_plain_c99():
    """include/*.h compile as C99 with -pedantic -Werror on their own (no C++ types, no extensions): the boundary a Go / Java / Rust host binds"""
    import subprocess
    for h in sorted(os.listdir(os.path.join(ROOT, "include"))):
        src = '#include "%s"\nint main(void) { return (int)sizeof(gk_opts) > 0 ? 0 : 1; }\n' % h if h == "gkgpu.h" else '#include "gkgpu.h"\n#include "%s"\nint main(void) { return 0; }\n' % h
        r = subprocess.run(["gcc", "-std=c99", "-Wall", "-Wextra", "-pedantic", "-Werror", "-fsyntax-only", "-I" + os.path.join(ROOT, "include"), "-x", "c", "-"],
                           input=src, capture_output=True, text=True)
        assert r.returncode == 0, (h, r.stderr[-2000:])


def test_c99_shim_walks_the_drivers_life_cycle_on_the_cpu_build(tmp_path):
    """The call order of INTEGRATION.md's cgo shim from plain C: engine with gk_opts (stats flag, disabled builtins), templates,
    constraints, data, 32 pthreads in gk_query_ex while another thread replaces the serving template and adds / removes a constraint
    (every answer is one template's or the other's, never a mixture; every input buffer is poisoned and freed right after its call:
    borrowed for the call only), the audit's table path, removal, shutdown.  Here against the TEST-ONLY CPU build of the engine."""
    import subprocess
    exe = _build_shim(tmp_path, L.library_path(True))
    r = subprocess.run([exe, "32", "40"], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0 and "abi_shim ok: 32 threads x 40 queries" in r.stdout, (r.stdout[-500:], r.stderr[-2000:])


@pytest.mark.gpu
def test_c99_shim_walks_the_drivers_life_cycle_on_the_device(tmp_path):
    """... and against libgkgpu.so on the MI355X: the same binary logic, the product library, no Python in the process"""
    import subprocess
    exe = _build_shim(tmp_path, L.library_path(False))
    env = dict(os.environ)
    import torch   # (the product library resolves libamdhip64 from the ROCm install when no torch is in the process; make sure the loader finds one)
    env["LD_LIBRARY_PATH"] = os.pathsep.join([os.path.join(os.path.dirname(torch.__file__), "lib"), "/opt/rocm/lib", env.get("LD_LIBRARY_PATH", "")])
    r = subprocess.run([exe, "32", "40"], capture_output=True, text=True, timeout=600, env=env)
    assert r.returncode == 0 and "abi_shim ok: 32 threads x 40 queries" in r.stdout, (r.stdout[-500:], r.stderr[-2000:])
