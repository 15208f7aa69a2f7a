"""The compiled "restated-reference CPU" loop (oracle/cpu_ref.cpp: the serial audit loop of pkg/audit/manager.go:591-642
with an independent C++ restatement of match.go) against the pure-Python oracle: same violating pairs, same autoreject
pairs, same result totals.  bench.py times this loop as `cpu_baseline` and checks the device bitmap against it."""
import json

import numpy as np
import pytest

import reference_tables as T
from gatekeeper_amd import _lib as L
from gatekeeper_amd import driver as D
from gatekeeper_amd import synth
from oracle import client as OC
from oracle import cpu_ref as CR
from oracle import target as OT


def _bits(row, n):
    return set(np.nonzero(np.unpackbits(row.view(np.uint8), bitorder="little")[:n])[0].tolist())


def _oracle_pairs(oc, cons, reviews):
    viol, err, results = {}, {}, {}
    for i, rv in enumerate(reviews):
        for r in oc.review(rv, OC.AUDIT_EP, None):
            k = (r.constraint.get("kind"), r.constraint["metadata"]["name"])
            if r.msg.startswith("unable to match constraints: "):
                err.setdefault(k, set()).add(i)
            else:
                viol.setdefault(k, set()).add(i)
                results[k] = results.get(k, 0) + 1
    return viol, err, results


@pytest.mark.parametrize("mixed", [False, True, "corpus"])
def test_cpu_ref_matches_python_oracle_on_synthetic(mixed, fixtures):
    n = 400
    templates = synth.psp_templates(fixtures)
    cons = synth.audit_constraints() if mixed else synth.psp_constraints()
    if mixed == "corpus":   # configs[4]'s 200 templates / constraints
        n, mixed = 150, True
        templates, cons = synth.corpus()
    oc = OC.Client()
    for t in templates:
        oc.add_template(t)
    for k in cons:
        oc.add_constraint(k)
    nss = synth.gen_namespaces()
    objs = synth.gen_objects(n, seed=41, mixed=mixed)
    want_v, want_e, want_r = _oracle_pairs(oc, cons, [OT.AugmentedUnstructured(OT.Unstructured(o), synth.namespace_for(o, nss), "Original") for o in objs])
    ref = CR.CpuRef(templates, [oc.constraints[(k["kind"], k["metadata"]["name"])][0] for k in cons])
    batch = synth.NativeBatch(L.load(hostemu=True), n, seed=41, mixed=mixed, namespaces=nss)
    for threads in (1, 3):
        out = ref.review(batch.reviews, n, threads)
        for row, key in enumerate(ref.keys):
            assert _bits(out["viol"][row], n) == want_v.get(key, set()), key
            assert _bits(out["err"][row], n) == want_e.get(key, set()), key
            assert int(out["results"][row]) == want_r.get(key, 0), key
        assert out["rejected"].sum() == 0 and out["seconds"] > 0
    assert sum(want_r.values()) > 0


def test_cpu_ref_match_layer_tables(fixtures):
    """match_test.go:17-684 (47 rows) and target_test.go:657-981 (TestMatcher_Match) through the C++ match restatement."""
    tmpl = fixtures["go_consts"]["pkg/target/target_integration_test.go"]["testTemplate"]["docs"][0]
    lib = L.load(hostemu=True)
    eng = D.Engine(hostemu=True)   # only to marshal gk_review_in arrays (Table creation is not used)

    def run(cons, rin):
        ref = CR.CpuRef([tmpl], [cons])
        arr = (L.gk_review_in * 1)()
        a = arr[0]
        a.kind, a.source, a.json, a.json_len, a.operation = rin.kind, rin.source, rin.json, len(rin.json), rin.operation
        if rin.namespace is not None:
            a.namespace_json, a.namespace_len = rin.namespace, len(rin.namespace)
        out = ref.review(arr, 1, 1)
        return bool(out["viol"][0][0] & 1), bool(out["err"][0][0] & 1)

    for name, obj, mt, ns, source, want, want_err in T.MATCH_CASES:
        if obj is None:
            continue
        cons = {"apiVersion": "constraints.gatekeeper.sh/v1beta1", "kind": "DenyAll", "metadata": {"name": "c"}, "spec": {"match": mt}}
        v, e = run(cons, D.to_review_in(D.AugmentedUnstructured(D.Unstructured(obj), ns, source)))
        assert e == bool(want_err), name
        if not want_err:
            assert v is want, name
    for name, shape, body, ns, cached, mt, want, want_err in T.MATCHER_MATCH_CASES:
        if cached:
            continue   # the loop takes the Namespace from the review (the audit shape); nsCache rows belong to the Client tests
        cons = {"apiVersion": "constraints.gatekeeper.sh/v1beta1", "kind": "DenyAll", "metadata": {"name": "c"}, "spec": {"match": mt}}
        rv = D.AugmentedUnstructured(D.Unstructured(body), ns, "") if shape == "object" else D.AugmentedReview(D.AdmissionRequest(dict(body)), ns, "")
        v, e = run(cons, D.to_review_in(rv))
        assert e == bool(want_err), name
        if not want_err:
            assert v is want, name
    del eng, lib
