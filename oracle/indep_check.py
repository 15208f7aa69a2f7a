"""ORACLE (test infrastructure only -- never imported by the product path).

ctypes binding of oracle/libgkindep.so (oracle/indep_check.cpp): the INDEPENDENT COMPILED CHECKER -- a C++ restatement of this
directory's Python oracle (own JSON reader, value model, Rego parser + interpreter, Match layer), built from that one file with
no object of the product on the link line.  It answers, for n object reviews given as JSON text, the violation and autoreject
bitmaps [n_constraints][ceil(n / 64)] -- the shape of the device's answer.

What stays in Python (this directory's own code, not the product's): the structural-schema defaulting of a constraint's
parameters (oracle/client.py apply_schema_defaults, what Client.AddConstraint does) and the check that no constraint asks for
scoped enforcement actions (the checker evaluates at the audit enforcement point only)."""
from __future__ import annotations

import ctypes as C
import json
import os

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))


class ReviewIn(C.Structure):
    """include/gkgpu.h gk_review_in: the public C struct (pointers and lengths) the batch's JSON text is handed over in"""
    _fields_ = [("kind", C.c_int32), ("source", C.c_int32), ("json", C.c_char_p), ("json_len", C.c_size_t),
                ("namespace_json", C.c_char_p), ("namespace_len", C.c_size_t), ("ns_object_json", C.c_char_p), ("ns_object_len", C.c_size_t),
                ("operation", C.c_char_p)]


def _load():
    path = os.path.join(_HERE, "libgkindep.so")
    if not os.path.exists(path):
        raise RuntimeError("oracle/libgkindep.so is not built (make -C oracle)")
    lib = C.CDLL(path)
    lib.ic_create.restype = C.c_void_p
    lib.ic_create.argtypes = [C.c_char_p, C.c_char_p]
    lib.ic_destroy.argtypes = [C.c_void_p]
    lib.ic_last_error.restype = C.c_char_p
    lib.ic_check.restype = C.c_int
    lib.ic_check.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.c_void_p, C.c_void_p, C.c_size_t, C.c_int]
    lib.ic_check_reviews.restype = C.c_int
    lib.ic_check_reviews.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.c_void_p, C.c_void_p, C.c_void_p, C.c_size_t, C.c_int]
    lib.ic_check_totals.restype = C.c_int
    lib.ic_check_totals.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_size_t, C.c_int]
    lib.ic_messages.restype = C.c_void_p
    lib.ic_messages.argtypes = [C.c_void_p, C.c_void_p]
    lib.ic_free.argtypes = [C.c_void_p]
    return lib


class IndepChecker:
    def __init__(self, templates, constraints):
        from . import client as OC
        self.lib = _load()
        by_kind = {OC.template_source(t)[0].lower(): t for t in templates}
        rows = []
        for c in constraints:
            if OC.get_enforcement_action(c) == "scoped":
                raise RuntimeError("scoped enforcement actions are outside the compiled checker's scope")
            rows.append(OC.apply_schema_defaults(by_kind[c.get("kind", "").lower()], c))
        self.n_constraints = len(rows)
        self.h = self.lib.ic_create(json.dumps(templates).encode(), json.dumps(rows).encode())
        if not self.h:
            raise RuntimeError("compiled checker: " + self.lib.ic_last_error().decode())

    def check(self, reviews, n, threads=None):
        """reviews: a ctypes array / pointer of gk_review_in (JSON text of bare objects + their Namespaces).  -> (viol, err)"""
        words = (n + 63) // 64
        viol = np.zeros((self.n_constraints, max(words, 1)), dtype=np.uint64)
        err = np.zeros_like(viol)
        threads = threads or max(1, min(os.cpu_count() or 1, 64))
        rc = self.lib.ic_check(self.h, C.cast(reviews, C.c_void_p), n, viol.ctypes.data, err.ctypes.data, max(words, 1), threads)
        if rc != 0:
            raise RuntimeError("compiled checker: " + self.lib.ic_last_error().decode())
        return viol, err

    def check_reviews(self, reviews, n, threads=None):
        """every shape of gk_review_in (bare objects with Operation / Source, AdmissionRequests, DELETE).  -> (viol, err, rejected):
        rejected[i] = 1 where HandleReview refuses review i (it has no results at all)"""
        words = (n + 63) // 64
        viol = np.zeros((self.n_constraints, max(words, 1)), dtype=np.uint64)
        err = np.zeros_like(viol)
        rejected = np.zeros(max(n, 1), dtype=np.uint8)
        threads = threads or max(1, min(os.cpu_count() or 1, 64))
        rc = self.lib.ic_check_reviews(self.h, C.cast(reviews, C.c_void_p), n, viol.ctypes.data, err.ctypes.data, rejected.ctypes.data, max(words, 1), threads)
        if rc != 0:
            raise RuntimeError("compiled checker: " + self.lib.ic_last_error().decode())
        return viol, err, rejected[:n]

    def check_totals(self, reviews, n, threads=None):
        """as check(), plus the RESULT totals per constraint (pkg/audit/manager.go:893-904: one per types.Result = per distinct
        (msg, details) of a violating pair).  -> (viol, err, results[n_constraints])"""
        words = (n + 63) // 64
        viol = np.zeros((self.n_constraints, max(words, 1)), dtype=np.uint64)
        err = np.zeros_like(viol)
        results = np.zeros(max(self.n_constraints, 1), dtype=np.uint64)
        threads = threads or max(1, min(os.cpu_count() or 1, 64))
        rc = self.lib.ic_check_totals(self.h, C.cast(reviews, C.c_void_p), n, viol.ctypes.data, err.ctypes.data, None, results.ctypes.data, max(words, 1), threads)
        if rc != 0:
            raise RuntimeError("compiled checker: " + self.lib.ic_last_error().decode())
        return viol, err, results[:self.n_constraints]

    def check_texts(self, texts, threads=None):
        """texts: [(object JSON text, namespace JSON text | None)]"""
        arr = (ReviewIn * max(len(texts), 1))()
        keep = []
        for i, (t, ns) in enumerate(texts):
            t = t.encode() if isinstance(t, str) else t
            ns = (ns.encode() if isinstance(ns, str) else ns) if ns else None
            keep.append((t, ns))
            arr[i].kind, arr[i].source, arr[i].json, arr[i].json_len = 1, 1, t, len(t)   # (bare object, Source "Original")
            arr[i].namespace_json, arr[i].namespace_len = ns, len(ns) if ns else 0
        return self.check(arr, len(texts), threads)

    def messages(self, text, ns=None):
        """the messages of ONE object: {row: [msg, ..]} -- one msg per distinct (msg, details), in the checker's set order"""
        r = ReviewIn()
        t = text.encode() if isinstance(text, str) else text
        nsb = (ns.encode() if isinstance(ns, str) else ns) if ns else None
        r.kind, r.source, r.json, r.json_len, r.namespace_json, r.namespace_len = 1, 1, t, len(t), nsb, len(nsb) if nsb else 0
        p = self.lib.ic_messages(self.h, C.byref(r))
        if not p:
            raise RuntimeError("compiled checker: " + self.lib.ic_last_error().decode())
        try:
            return {int(k): v for k, v in json.loads(C.string_at(p).decode()).items()}
        finally:
            self.lib.ic_free(p)

    def close(self):
        if self.h:
            self.lib.ic_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:   # noqa: BLE001
            pass
