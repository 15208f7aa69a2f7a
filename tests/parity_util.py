"""Shared helpers for the parity tests: the product path (gatekeeper_amd, through the C ABI) against the oracle."""
import json
import os

import pytest

from gatekeeper_amd import driver as D
from gatekeeper_amd import synth
from oracle import client as OC
from oracle import target as OT

BACKENDS = [pytest.param("hostemu", id="hostemu"), pytest.param("gpu", marks=pytest.mark.gpu, id="gpu")]


def make_client(backend, **kw):
    """backend 'gpu': the product library (HIP kernels on cuda:0).  backend 'hostemu': the TEST-ONLY build in which
    the same vm_core.hpp code runs lane by lane on the CPU (GPU-less container)."""
    drv = D.Driver(hostemu=(backend == "hostemu"), **kw)
    return D.Client(drv)


def key(r):
    c = r.constraint
    return (c.get("kind"), (c.get("metadata") or {}).get("name"), r.msg, json.dumps(r.metadata, sort_keys=True),
            r.enforcement_action, tuple(r.scoped_enforcement_actions or ()))


def to_oracle_review(obj):
    if isinstance(obj, D.AugmentedUnstructured):
        return OT.AugmentedUnstructured(OT.Unstructured(obj.object), obj.namespace, obj.source, obj.operation)
    if isinstance(obj, D.AugmentedReview):
        return OT.AugmentedReview(OT.AdmissionRequest(obj.admission_request), obj.namespace, obj.source, obj.is_admission)
    if isinstance(obj, D.AdmissionRequest):
        return OT.AdmissionRequest(obj)
    if isinstance(obj, D.Unstructured):
        return OT.Unstructured(obj)
    return obj


def load_both(backend, templates, constraints, data=(), **kw):
    c = make_client(backend, **kw)
    oc = OC.Client()
    for t in templates:
        c.AddTemplate(t)
        oc.add_template(t)
    for k in constraints:
        c.AddConstraint(k)
        oc.add_constraint(k)
    for d in data:
        c.AddData(d)
        oc.add_data(d)
    return c, oc


def assert_parity(c, oc, reviews, ep=D.AUDIT_EP, namespaces=None):
    got = c.ReviewBatch(reviews, ep, namespaces)
    total = 0
    for i, (rv, g) in enumerate(zip(reviews, got)):
        exp = oc.review(to_oracle_review(rv), ep, namespaces[i] if namespaces else None)
        a, b = sorted(key(r) for r in g), sorted(key(r) for r in exp)
        assert a == b, "review %d: device %r != oracle %r" % (i, a, b)
        total += len(b)
    return total
