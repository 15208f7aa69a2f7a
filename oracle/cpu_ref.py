"""ORACLE / CPU BASELINE (test infrastructure only): ctypes wrapper of oracle/libgkcpuref.so -- the compiled
"restated-reference CPU" loop of oracle/cpu_ref.cpp (the reference's serial audit loop, pkg/audit/manager.go:591-642,
in C++).  Used by tests/test_cpu_ref.py (pinned against the pure-Python oracle) and by bench.py's cpu_baseline leg."""
from __future__ import annotations

import ctypes as C
import json
import os

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_lib = None


def load():
    global _lib
    if _lib is None:
        path = os.path.join(_HERE, "libgkcpuref.so")
        if not os.path.exists(path):
            raise RuntimeError("%s not found: build it with `make -C oracle` (or __graft_entry__.build())" % path)
        lib = C.CDLL(path)
        lib.cpuref_create.restype = C.c_void_p
        lib.cpuref_destroy.argtypes = [C.c_void_p]
        lib.cpuref_last_error.restype = C.c_char_p
        lib.cpuref_add_template.argtypes = [C.c_void_p, C.c_char_p, C.c_char_p, C.POINTER(C.c_char_p), C.c_size_t]
        lib.cpuref_add_constraint.argtypes = [C.c_void_p, C.c_char_p, C.c_size_t]
        lib.cpuref_review.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p,
                                      C.POINTER(C.c_double)]
        _lib = lib
    return _lib


class CpuRef:
    """templates: ConstraintTemplate dicts; constraints: constraint dicts AFTER CRD defaulting (bitmap rows follow their order)."""

    def __init__(self, templates, constraints):
        from .client import template_source
        self.lib = load()
        self.h = C.c_void_p(self.lib.cpuref_create())
        for t in templates:
            kind, _, rego, libs = template_source(t)
            arr = (C.c_char_p * max(1, len(libs)))(*[x.encode() for x in libs])
            self._ok(self.lib.cpuref_add_template(self.h, kind.encode(), rego.encode(), arr, len(libs)))
        self.keys = []
        for c in constraints:
            body = json.dumps(c).encode()
            self._ok(self.lib.cpuref_add_constraint(self.h, body, len(body)))
            self.keys.append((c.get("kind", ""), (c.get("metadata") or {}).get("name", "")))

    def _ok(self, rc):
        if rc != 0:
            raise RuntimeError(self.lib.cpuref_last_error().decode())

    def review(self, reviews_ptr, n, threads=1):
        """reviews_ptr: a gk_review_in array (ctypes).  -> dict(viol, err: [C][ceil(n/64)] uint64, results: [C], rejected: [n], seconds)"""
        nc, nt = len(self.keys), (n + 63) // 64
        viol = np.zeros((nc, nt), np.uint64)
        err = np.zeros((nc, nt), np.uint64)
        results = np.zeros(nc, np.uint64)
        rejected = np.zeros(max(n, 1), np.uint8)
        secs = C.c_double()
        self._ok(self.lib.cpuref_review(self.h, C.cast(reviews_ptr, C.c_void_p), n, threads, viol.ctypes.data, err.ctypes.data,
                                        results.ctypes.data, rejected.ctypes.data, C.byref(secs)))
        return {"viol": viol, "err": err, "results": results, "rejected": rejected[:n], "seconds": secs.value}

    def close(self):
        if self.h:
            self.lib.cpuref_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass
