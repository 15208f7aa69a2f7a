"""ORACLE (test infrastructure only): the INDEPENDENT full-size parity leg of bench.py.

The device bitmaps of the timed table are compared with the pure-Python oracle (oracle/client.py: its own JSON reader --
Python's json --, its own HandleReview, Match layer and tree-walking Rego interpreter; it shares no code with the product)
on a sample of the SAME objects.  The objects are taken as JSON TEXT from the batch the table was built from, so nothing of
the product's parser or flattener sits between the bytes and the checker.  Worker processes (plain child interpreters) only partition the
objects; each builds its own oracle client."""
from __future__ import annotations

import json
import os
import time

_oc = None
_keys = None


def _init(templates_json, constraints_json):
    global _oc, _keys
    from oracle import client as OC
    _oc = OC.Client()
    for t in json.loads(templates_json):
        _oc.add_template(t)
    cons = json.loads(constraints_json)
    for c in cons:
        _oc.add_constraint(c)
    _keys = {(c["kind"], c["metadata"]["name"]): i for i, c in enumerate(cons)}


def _chunk(args):
    """[(index, object JSON text, namespace JSON text or None)] -> [(index, [violating constraint rows], [autoreject rows])]"""
    from oracle import client as OC
    from oracle import target as OT
    out = []
    for i, text, ns_text in args:
        obj = json.loads(text)
        ns = json.loads(ns_text) if ns_text else None
        viol, err = set(), set()
        for r in _oc.review(OT.AugmentedUnstructured(OT.Unstructured(obj), ns, "Original"), OC.AUDIT_EP):
            row = _keys[(r.constraint["kind"], r.constraint["metadata"]["name"])]
            (err if r.msg.startswith("unable to match constraints: ") and not r.metadata.get("details") else viol).add(row)
        out.append((i, sorted(viol), sorted(err)))
    return out


def python_oracle_pairs(templates, constraints, texts, procs=None, timeout_s=600):
    """texts: [(object JSON text, namespace JSON text | None)].  -> (viol pairs {(row, i)}, err pairs {(row, i)}, seconds, processes).
    Rows index `constraints`.  The workers are plain child interpreters (`python -m oracle.bench_leg in out`): nothing of the
    calling process -- which holds a HIP context -- is forked or re-imported."""
    import subprocess
    import sys
    import tempfile
    n = len(texts)
    procs = max(1, min(procs or (os.cpu_count() or 1), 64, (n + 511) // 512))
    items = [(i, t.decode() if isinstance(t, bytes) else t, (ns.decode() if isinstance(ns, bytes) else ns)) for i, (t, ns) in enumerate(texts)]
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    t0 = time.perf_counter()
    viol, err = set(), set()
    with tempfile.TemporaryDirectory() as tmp:
        kids = []
        for w in range(procs):
            part = items[w::procs]
            fin, fout = os.path.join(tmp, "in_%d.json" % w), os.path.join(tmp, "out_%d.json" % w)
            with open(fin, "w") as fh:
                json.dump({"templates": templates, "constraints": constraints, "items": part}, fh)
            kids.append((subprocess.Popen([sys.executable, "-m", "oracle.bench_leg", fin, fout], cwd=root), fout))
        for kid, fout in kids:
            rc = kid.wait(timeout=timeout_s)
            if rc != 0:
                raise RuntimeError("python-oracle worker failed with exit status %d" % rc)
            for i, v, e in json.load(open(fout)):
                viol.update((row, i) for row in v)
                err.update((row, i) for row in e)
    return viol, err, time.perf_counter() - t0, procs


if __name__ == "__main__":
    import sys
    job = json.load(open(sys.argv[1]))
    _init(json.dumps(job["templates"]), json.dumps(job["constraints"]))
    with open(sys.argv[2], "w") as fh:
        json.dump(_chunk([tuple(x) for x in job["items"]]), fh)
