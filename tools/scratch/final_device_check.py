#!/usr/bin/env python3
"""A torch-free device check for the tail of a round's GPU budget: the PSP set over 256 pods plus the message-text templates of
tests/test_sprintf_matrix.py through the C ABI on the device, compared with the Python oracle.  ~15 s on a fresh box."""
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
t0 = time.time()
from gatekeeper_amd import driver as D   # noqa: E402
from gatekeeper_amd import synth          # noqa: E402
from oracle import client as OC           # noqa: E402
from oracle import target as OT           # noqa: E402
import test_sprintf_matrix as M           # noqa: E402

fx = synth.load_fixtures()
c, oc = D.Client(D.Driver(device=0, hostemu=os.environ.get("GK_CHECK_ON_CPU_BUILD") == "1")), OC.Client()   # (the env switch: a dry run of this script in the build container)
cons, _n = M._constraint("vsdqxXoObcfFeEgGtTUp")
extra_t = [M.TEMPLATE, M.QUOTE_TEMPLATE]
extra_c = [cons, {"apiVersion": "constraints.gatekeeper.sh/v1beta1", "kind": "K8sQuote", "metadata": {"name": "q"}, "spec": {}}]
for t in synth.psp_templates(fx) + extra_t:
    c.AddTemplate(t)
    oc.add_template(t)
for k in synth.psp_constraints() + extra_c:
    c.AddConstraint(k)
    oc.add_constraint(k)
nss = synth.gen_namespaces()
objs = synth.gen_objects(256, seed=3) + M._objs() + [{"apiVersion": "v1", "kind": "Pod", "metadata": {"name": "cp%d" % i}, "spec": {"x": "a" + chr(cp) + "z"}} for i, cp in enumerate(M.CODE_POINTS)]
rv = [D.AugmentedUnstructured(D.Unstructured(o), synth.namespace_for(o, nss) if o["metadata"].get("namespace") else None, "Original") for o in objs]
got = c.ReviewBatch(rv, D.AUDIT_EP)
n = 0
for o, g in zip(objs, got):
    ns = synth.namespace_for(o, nss) if o["metadata"].get("namespace") else None
    exp = oc.review(OT.AugmentedUnstructured(OT.Unstructured(o), ns, "Original"), OC.AUDIT_EP)
    a = sorted((r.constraint["metadata"]["name"], r.msg) for r in g)
    b = sorted((r.constraint["metadata"]["name"], r.msg) for r in exp)
    assert a == b, (o["metadata"]["name"], [x for x in a if x not in b][:2], [x for x in b if x not in a][:2])
    n += len(b)
print(json.dumps({"device_check": "ok", "objects": len(objs), "results": n, "seconds": round(time.time() - t0, 1)}))
