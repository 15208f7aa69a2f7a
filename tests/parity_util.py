"""Shared helpers for the parity tests: the product path (gatekeeper_amd, through the C ABI) against the oracle."""
import json
import os

import pytest

from gatekeeper_amd import driver as D
from gatekeeper_amd import synth
from oracle import client as OC
from oracle import target as OT

BACKENDS = [pytest.param("hostemu", id="hostemu"), pytest.param("hostemu-gen", id="hostemu-gen"),
            pytest.param("gpu", marks=pytest.mark.gpu, id="gpu"), pytest.param("gpu-interp", marks=pytest.mark.gpu, id="gpu-interp")]


def make_client(backend, **kw):
    """backend 'gpu': the product library on cuda:0 with the plan-specialised (hiprtc) kernel; 'gpu-interp': the same
    library forced onto its generic bytecode kernel (GK_NO_JIT=1).  'hostemu' / 'hostemu-gen': the TEST-ONLY build in
    which the interpreter / the generated plan source (compiled with g++) runs lane by lane on the CPU."""
    os.environ.pop("GK_NO_JIT", None)
    os.environ.pop("GK_HOSTEMU_JIT", None)
    os.environ.pop("GK_JIT_STRICT", None)
    if backend == "gpu-interp":
        os.environ["GK_NO_JIT"] = "1"
    if backend == "gpu":
        os.environ["GK_JIT_STRICT"] = "1"   # a hiprtc failure must fail the test, not silently use the interpreter
    if backend == "hostemu-gen":
        os.environ["GK_HOSTEMU_JIT"] = "1"
    drv = D.Driver(hostemu=backend.startswith("hostemu"), **kw)
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


def load_both(backend, templates, constraints, data=(), validate=True, **kw):
    """validate=False installs constraints without the target handler's ValidateConstraint (Match-layer error-path tests)"""
    c = make_client(backend, **kw)
    oc = OC.Client()
    for t in templates:
        c.AddTemplate(t)
        oc.add_template(t)
    for k in constraints:
        c.AddConstraint(k, validate=validate)
        oc.add_constraint(k, validate=validate)
    for d in data:
        c.AddData(d)
        oc.add_data(d)
    return c, oc


def object_where_elements_are_iterated(review):
    """True when some container-ish member of the reviewed object(s) is a non-empty OBJECT under a key where Kubernetes
    holds arrays (the paths the PSP templates iterate with `[_]`): the engine refuses such reviews (fail closed) because
    Rego would walk the object's values."""
    arrays = ("containers", "initContainers", "ephemeralContainers", "volumes", "ports", "volumeMounts", "env")

    def walk(v):
        if isinstance(v, dict):
            return any((k in arrays and isinstance(x, dict) and x) or walk(x) for k, x in v.items())
        if isinstance(v, list):
            return any(walk(x) for x in v)
        return False
    body = review.object if isinstance(review, D.AugmentedUnstructured) else review.admission_request if isinstance(review, D.AugmentedReview) else review
    return walk(body) or compared_value_is_a_container(body)


def compared_value_is_a_container(body):
    """True when a member the loaded templates compare with another review value (`volumeMounts[_].name == volumes[_].name`)
    holds a NON-EMPTY container: equality of two such values is a deep comparison the plan cannot make, so the engine
    refuses the review (fail closed) instead of comparing payloads."""
    def walk(v):
        if isinstance(v, dict):
            return any((k == "name" and isinstance(x, (dict, list)) and len(x) > 0) or walk(x) for k, x in v.items())
        if isinstance(v, list):
            return any(walk(x) for x in v)
        return False
    return walk(body)


def assert_parity(c, oc, reviews, ep=D.AUDIT_EP, namespaces=None, refused=None):
    """refused (a list, or None): reviews the engine REFUSES with a LimitError are collected there instead of failing the
    comparison, provided they hold an object where array elements are iterated or a non-empty container where two review
    values are compared; everything else must match.
    Product == oracle for every review: (1) the rendered result multisets (constraint, msg, details, actions), and
    (2) the RAW device bitmaps -- the violation bits and the match-error bits, before any host rendering -- against the
    pair sets the oracle's results imply, so that a spurious device bit cannot hide behind a renderer that returns []."""
    got = c.ReviewBatch(reviews, ep, namespaces)
    total = 0
    want_viol, want_err = set(), set()
    for i, (rv, g) in enumerate(zip(reviews, got)):
        exp = oc.review(to_oracle_review(rv), ep, namespaces[i] if namespaces else None)
        if refused is not None and isinstance(g, D.ReviewFailure) and isinstance(g.cause, D.LimitError):
            assert object_where_elements_are_iterated(rv), "review %d refused without cause: %r" % (i, g)
            refused.append(i)
            continue
        assert not isinstance(g, Exception), "review %d: %r" % (i, g)
        a, b = sorted(key(r) for r in g), sorted(key(r) for r in exp)
        assert a == b, "review %d: device %r != oracle %r" % (i, a, b)
        total += len(b)
        for r in exp:
            k = (r.constraint.get("kind"), (r.constraint.get("metadata") or {}).get("name"))
            (want_err if r.msg.startswith("unable to match constraints: ") and not r.metadata.get("details") else want_viol).add((k, i))
    rins = [D.to_review_in(rv, namespaces[i] if namespaces else None) for i, rv in enumerate(reviews)]
    table = c.driver.engine.create_table(rins, keep_docs=False)
    try:
        ev = table.eval()
        active = {cid: (cons.get("kind"), (cons.get("metadata") or {}).get("name")) for cid, (cons, _, _) in c._active(ep).items()}
        skip = set(refused or ())
        assert set(int(r) for r in ev.too_big_reviews()) == skip
        dev_viol = {(active[cid], r) for cid, r in ev.pairs("viol") if cid in active and r not in skip}
        dev_err = {(active[cid], r) for cid, r in ev.pairs("err") if cid in active and r not in skip}
        assert dev_viol == want_viol, "raw violation bitmap != oracle pairs: only device %r, only oracle %r" % (
            sorted(dev_viol - want_viol)[:5], sorted(want_viol - dev_viol)[:5])
        assert dev_err == want_err, "raw match-error bitmap != oracle pairs: only device %r, only oracle %r" % (
            sorted(dev_err - want_err)[:5], sorted(want_err - dev_err)[:5])
    finally:
        table.free()
    return total


import contextlib   # noqa: E402


@contextlib.contextmanager
def plan_group_max(lib, n):
    """gk_debug_set("group_max", n) for the policy changes inside the block: plan groups of at most n constraints -- what rounds 1-5 did
    at 64; since round 6 a plan holds up to 256 distinct violation formulas and the in-tree corpora are ONE plan.  The several-group path
    (still taken beyond 256 violation / 64 match formulas) stays tested through this knob."""
    assert lib.gk_debug_set(b"group_max", int(n)) == 0
    try:
        yield
    finally:
        lib.gk_debug_set(b"group_max", 0)
