"""The independent compiled checker (oracle/indep_check.cpp -> oracle/libgkindep.so: no product object linked) against the
Python oracle it restates: the same violation and autoreject pairs on the bench's policy sets x synthetic objects, structurally
mutated objects and objects without namespaces / with broken labels."""
import json
import subprocess
import os

import numpy as np
import pytest

from gatekeeper_amd import synth
from oracle import client as OC
from oracle import target as OT
from oracle.indep_check import IndepChecker

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_the_checker_links_nothing_of_the_product():
    subprocess.run(["make", "-C", os.path.join(ROOT, "oracle"), "libgkindep.so"], check=True, capture_output=True, timeout=600)
    mk = open(os.path.join(ROOT, "oracle", "Makefile")).read()
    rule = [l for l in mk.splitlines() if "indep_check.cpp" in l and "$(CXX)" in l]
    assert rule and "$(OBJS)" not in rule[0] and "build/" not in rule[0] and "gkgpu" not in rule[0].replace("libgkindep", "")
    out = subprocess.run(["ldd", os.path.join(ROOT, "oracle", "libgkindep.so")], capture_output=True, text=True).stdout
    assert "gkgpu" not in out and "hip" not in out.lower() and "torch" not in out
    src = open(os.path.join(ROOT, "oracle", "indep_check.cpp")).read()
    assert [l for l in src.splitlines() if l.startswith("#include \"")] == ['#include "../include/gkgpu.h"   // gk_review_in only (plain C struct: pointers and lengths)']


def _python_pairs(templates, constraints, objs_ns):
    oc = OC.Client()
    for t in templates:
        oc.add_template(t)
    for c in constraints:
        oc.add_constraint(c)
    keys = {(c["kind"], c["metadata"]["name"]): i for i, c in enumerate(constraints)}
    viol, err = set(), set()
    for i, (o, ns) in enumerate(objs_ns):
        for r in oc.review(OT.AugmentedUnstructured(OT.Unstructured(o), ns, "Original"), OC.AUDIT_EP):
            row = keys[(r.constraint["kind"], r.constraint["metadata"]["name"])]
            (err if r.msg.startswith("unable to match constraints: ") and not r.metadata.get("details") else viol).add((row, i))
    return viol, err


def _pairs_of(bm, n):
    out = set()
    for row in range(bm.shape[0]):
        bits = np.unpackbits(bm[row].view(np.uint8), bitorder="little")[:n]
        out.update((row, int(i)) for i in np.nonzero(bits)[0])
    return out


@pytest.mark.parametrize("policy", ["audit-50", "psp-30", "corpus"])
def test_compiled_checker_equals_the_python_oracle(policy, fixtures):
    if policy == "audit-50":
        ts, cs = synth.psp_templates(fixtures), synth.audit_constraints()
    elif policy == "psp-30":
        ts, cs = synth.psp_templates(fixtures), synth.psp_constraints()
    else:
        ts, cs = synth.corpus(fixtures)
        ts = ts[::3]
        kinds = {t["spec"]["crd"]["spec"]["names"]["kind"] for t in ts}
        cs = [c for c in cs if c["kind"] in kinds]
    nss = synth.gen_namespaces()
    objs = synth.gen_objects(260, seed=41, mixed=True)
    objs_ns = [(o, synth.namespace_for(o, nss)) for o in objs]
    # objects the reference's decoder / matcher trips over: no kind, labels that are no string map, no namespace, a Namespace
    objs_ns += [({"apiVersion": "v1", "metadata": {"name": "nokind"}}, None),
                ({"apiVersion": "v1", "kind": "Pod", "metadata": {"name": "badlabels", "namespace": "dev-00", "labels": {"a": 1}}, "spec": {"containers": [{"name": "c", "image": "nginx"}]}}, None),
                ({"apiVersion": "v1", "kind": "Pod", "metadata": {"name": "nons"}, "spec": {"containers": [{"name": "c", "image": "nginx", "securityContext": {"privileged": True}}]}}, None),
                ({"apiVersion": "v1", "kind": "Namespace", "metadata": {"name": "prod-00", "labels": {"env": "prod"}}}, None),
                ({"apiVersion": "apps/v1", "kind": "Deployment", "metadata": {"name": "d", "namespace": "dev-01"}, "spec": {"template": {"spec": {"containers": "notalist"}}}}, synth.gen_namespaces().get("dev-01"))]
    want_v, want_e = _python_pairs(ts, cs, objs_ns)
    ck = IndepChecker(ts, cs)
    viol, err = ck.check_texts([(json.dumps(o), json.dumps(ns) if ns is not None else None) for o, ns in objs_ns], threads=3)
    got_v, got_e = _pairs_of(viol, len(objs_ns)), _pairs_of(err, len(objs_ns))
    assert len(want_v) > 100
    assert got_v == want_v, (sorted(got_v - want_v)[:5], sorted(want_v - got_v)[:5])
    assert got_e == want_e, (sorted(got_e - want_e)[:5], sorted(want_e - got_e)[:5])


def test_compiled_checker_on_structurally_mutated_objects(fixtures):
    """wrong types, missing members, arrays where objects are expected (tests/test_parity.py _mutate): both restatements answer alike"""
    from test_parity import _mutate
    ts, cs = synth.psp_templates(fixtures), synth.audit_constraints()
    nss = synth.gen_namespaces()
    objs_ns = []
    for seed in (5, 6, 7):
        rng = synth.SplitMix64(seed)
        for o in synth.gen_objects(120, seed=seed, mixed=True):
            m = _mutate(rng, _mutate(rng, o))
            if not isinstance(m, dict):
                m = {"apiVersion": "v1", "kind": "Pod", "metadata": {"name": "m"}}
            md = m.get("metadata")
            ns = synth.namespace_for(m, nss) if isinstance(md, dict) and isinstance(md.get("namespace"), str) else None
            objs_ns.append((m, ns))
    want_v, want_e = _python_pairs(ts, cs, objs_ns)
    ck = IndepChecker(ts, cs)
    viol, err = ck.check_texts([(json.dumps(o), json.dumps(ns) if ns is not None else None) for o, ns in objs_ns], threads=2)
    got_v, got_e = _pairs_of(viol, len(objs_ns)), _pairs_of(err, len(objs_ns))
    assert len(want_v) > 100 and len(want_e) > 0
    assert got_v == want_v, (sorted(got_v - want_v)[:5], sorted(want_v - got_v)[:5])
    assert got_e == want_e, (sorted(got_e - want_e)[:5], sorted(want_e - got_e)[:5])


@pytest.mark.parametrize("policy,n", [("audit-50", 20000), ("corpus-200", 3000)])
def test_product_equals_the_compiled_checker_at_sizes_the_python_oracle_does_not_reach(policy, n, fixtures):
    """the product's bitmaps (CPU build of the engine here; the gpu suite and bench.py do the same on the device) against the independent
    compiled checker on every object -- tens of thousands of objects x the whole policy set in seconds"""
    from gatekeeper_amd import driver as D
    ts, cs = (synth.psp_templates(fixtures), synth.audit_constraints()) if policy == "audit-50" else synth.corpus(fixtures)
    drv = D.Driver(device=0, hostemu=True)
    client = D.Client(drv)
    for t in ts:
        client.AddTemplate(t)
    for c in cs:
        client.AddConstraint(c)
    batch = synth.NativeBatch(drv.engine.lib, n, seed=synth.SEED + 3, mixed=True, start=0, namespaces=synth.gen_namespaces())
    table = drv.engine.create_table_native(batch.reviews, n, keep_docs=False, resident=True, pruned=True)
    table.launch()
    ev = table.eval(download=True, collect_only=True)      # (the bench's shape: enqueue, then collect)
    ids = [drv.constraint_id(client.constraints[(k["kind"], k["metadata"]["name"])]) for k in cs]
    ck = IndepChecker(ts, cs)
    viol, err = ck.check(batch.reviews, n, threads=4)
    row_of = {int(cid): i for i, cid in enumerate(ev.constraint_ids)}
    words = (n + 63) // 64
    tail = np.uint64((1 << (n % 64)) - 1) if n % 64 else None
    pairs = 0
    for row, cid in enumerate(ids):
        d_v, d_e = np.array(ev.viol[row_of[cid]][:words], copy=True), np.array(ev.err[row_of[cid]][:words], copy=True)
        if tail is not None:
            d_v[-1] &= tail
            d_e[-1] &= tail
        assert (d_v == viol[row]).all() and (d_e == err[row]).all(), (policy, cs[row]["kind"], cs[row]["metadata"]["name"])
        pairs += int(np.unpackbits(d_v.view(np.uint8)).sum())
    assert pairs > n // 2
    assert len(ev.too_big_reviews()) == 0


def _python_messages(templates, constraints, objs_ns):
    oc = OC.Client()
    for t in templates:
        oc.add_template(t)
    for c in constraints:
        oc.add_constraint(c)
    keys = {(c["kind"], c["metadata"]["name"]): i for i, c in enumerate(constraints)}
    out = {}
    for i, (o, ns) in enumerate(objs_ns):
        for r in oc.review(OT.AugmentedUnstructured(OT.Unstructured(o), ns, "Original"), OC.AUDIT_EP):
            if r.msg.startswith("unable to match constraints: ") and not r.metadata.get("details"):
                continue
            out.setdefault((keys[(r.constraint["kind"], r.constraint["metadata"]["name"])], i), []).append(r.msg)
    return {k: sorted(v) for k, v in out.items()}


@pytest.mark.parametrize("policy", ["audit-50", "corpus"])
def test_compiled_checker_messages_equal_the_python_oracle(policy, fixtures):
    """the TEXT of every violation (sprintf verbs, number and composite formatting, concat/format_int), not only which pairs violate"""
    if policy == "audit-50":
        ts, cs = synth.psp_templates(fixtures), synth.audit_constraints()
    else:
        ts, cs = synth.corpus(fixtures)
        ts = ts[1::3]
        kinds = {t["spec"]["crd"]["spec"]["names"]["kind"] for t in ts}
        cs = [c for c in cs if c["kind"] in kinds]
    nss = synth.gen_namespaces()
    objs = synth.gen_objects(200, seed=43, mixed=True)
    # numbers the formatting rules differ on: integral floats, exponents, fractions, big integers
    objs += [{"apiVersion": "v1", "kind": "Pod", "metadata": {"name": "nums", "namespace": "dev-00"},
              "spec": {"containers": [{"name": "c", "image": "nginx", "securityContext": {"runAsUser": v, "privileged": True},
                                       "ports": [{"hostPort": v, "containerPort": 80}]}],
                       "securityContext": {"fsGroup": v, "supplementalGroups": [v, 1]}}}
             for v in (1.5, 2.0, 1e21, 1e-7, 123456789012, 0.000123, -3.25, 1e6, 999999.5, 1234567.25)]
    objs_ns = [(o, synth.namespace_for(o, nss)) for o in objs]
    want = _python_messages(ts, cs, objs_ns)
    ck = IndepChecker(ts, cs)
    got = {}
    for i, (o, ns) in enumerate(objs_ns):
        for row, msgs in ck.messages(json.dumps(o), json.dumps(ns) if ns is not None else None).items():
            got[(row, i)] = sorted(msgs)
    assert len(want) > 100
    assert got.keys() == want.keys()
    bad = [(k, got[k], want[k]) for k in want if got[k] != want[k]]
    assert not bad, bad[:3]


@pytest.mark.parametrize("policy,n", [("audit-50", 4000), ("corpus-200", 600)])
def test_product_messages_equal_the_compiled_checker(policy, n, fixtures):
    """every message the product renders (gk_render over the flagged pairs: the concrete evaluator, cross-checked against the partial
    evaluator by GK_RENDER_CHECK) against the independent compiled checker's text for the same pair"""
    from gatekeeper_amd import driver as D
    ts, cs = (synth.psp_templates(fixtures), synth.audit_constraints()) if policy == "audit-50" else synth.corpus(fixtures)
    drv = D.Driver(device=0, hostemu=True)
    client = D.Client(drv)
    for t in ts:
        client.AddTemplate(t)
    for c in cs:
        client.AddConstraint(c)
    batch = synth.NativeBatch(drv.engine.lib, n, seed=synth.SEED + 9, mixed=True, start=0, namespaces=synth.gen_namespaces())
    table = drv.engine.create_table_native(batch.reviews, n, keep_docs=True, resident=False, pruned=False)
    ev = table.eval()
    ids = [drv.constraint_id(client.constraints[(k["kind"], k["metadata"]["name"])]) for k in cs]
    row_of_cid = {cid: row for row, cid in enumerate(ids)}
    product = {}
    for cid, r in ev.pairs("viol"):
        product[(row_of_cid[int(cid)], int(r))] = sorted(v["msg"] for v in table.render(cid, r))
    ck = IndepChecker(ts, cs)
    checker = {}
    for i in range(n):
        rin = batch.reviews[i]
        for row, msgs in ck.messages(rin.json, rin.namespace_json).items():
            checker[(row, i)] = sorted(msgs)
    assert len(product) > n
    assert product.keys() == checker.keys()
    bad = [(k, product[k], checker[k]) for k in checker if product[k] != checker[k]]
    assert not bad, bad[:3]
    table.free()


FORMAT_TEMPLATE = {
    "apiVersion": "templates.gatekeeper.sh/v1", "kind": "ConstraintTemplate", "metadata": {"name": "k8sformats"},
    "spec": {"crd": {"spec": {"names": {"kind": "K8sFormats"}}},
             "targets": [{"target": "admission.k8s.gatekeeper.sh", "rego": """
package k8sformats
violation[{"msg": msg}] {
  x := input.review.object.spec.x
  msg := sprintf("v=%v|w=%8v|l=%-8v|z=%08v|plus=%+v|s=%s|d=%d|q=%q|all=%v|obj=%v|pct=%%|%v", [x, x, x, x, x, x, x, x, input.review.object.spec, {"k": x, "n": [x, 1.5]}])
}
violation[{"msg": msg}] {
  x := input.review.object.spec.x
  msg := sprintf("extra %v", [x, x, "tail"])
}
violation[{"msg": msg}] {
  x := input.review.object.spec.x
  msg := concat("/", [sprintf("%v", [x]), sprintf("%d", [input.review.object.spec.n]), sprintf("%5.2s|%v", ["héllo", [x]])])
}
"""}]}}


def test_message_formatting_three_ways():
    """sprintf verbs / flags / widths, MISSING and EXTRA operands, numbers in every notation and composite operands: the Python oracle,
    the compiled checker and the product's renderer print the same text"""
    from gatekeeper_amd import driver as D
    ts = [FORMAT_TEMPLATE]
    cs = [{"apiVersion": "constraints.gatekeeper.sh/v1beta1", "kind": "K8sFormats", "metadata": {"name": "f"}, "spec": {}}]
    xs = [1, -7, 2.0, 1.5, -3.25, 1e21, 1e20, 1e-7, 0.000123, 123456789012, 1e6, 999999.5, 1234567.25, 100000.0, 1e-5, 0.0001,
          "str", "", "with \"quote\" and \\ and é", True, None, [1, 2.5, "a"], {"a": 1e7, "b": [True, None]}, 9007199254740993, 0.1, 5e-324, 1.7976931348623157e308]
    objs_ns = [({"apiVersion": "v1", "kind": "Pod", "metadata": {"name": "p%d" % i}, "spec": {"x": x, "n": 255 + i}}, None) for i, x in enumerate(xs)]
    want = _python_messages(ts, cs, objs_ns)
    ck = IndepChecker(ts, cs)
    got = {}
    for i, (o, _ns) in enumerate(objs_ns):
        for row, msgs in ck.messages(json.dumps(o)).items():
            got[(row, i)] = sorted(msgs)
    assert len(want) == len(xs) and all(len(v) == 3 for v in want.values())
    bad = [(k, got.get(k), want[k]) for k in want if got.get(k) != want[k]]
    assert not bad, bad[:2]
    drv = D.Driver(device=0, hostemu=True)
    client = D.Client(drv)
    client.AddTemplate(ts[0])
    client.AddConstraint(cs[0])
    res = client.ReviewBatch([D.AugmentedUnstructured(D.Unstructured(o), None, "Original") for o, _ in objs_ns])
    product = {(0, i): sorted(r.msg for r in rs) for i, rs in enumerate(res)}
    bad = [(k, product.get(k), want[k]) for k in want if product.get(k) != want[k]]
    assert not bad, bad[:2]


ENVELOPE_TEMPLATE = {
    "apiVersion": "templates.gatekeeper.sh/v1", "kind": "ConstraintTemplate", "metadata": {"name": "k8senvelope"},
    "spec": {"crd": {"spec": {"names": {"kind": "K8sEnvelope"}}},
             "targets": [{"target": "admission.k8s.gatekeeper.sh", "rego": """
package k8senvelope
violation[{"msg": msg}] {
  input.review.operation == "DELETE"
  msg := sprintf("delete of %v by %v", [input.review.name, input.review.userInfo.username])
}
violation[{"msg": msg}] {
  input.review.dryRun == true
  msg := "dry run"
}
violation[{"msg": msg}] {
  input.review.oldObject.metadata.labels.was == "old"
  msg := "was old"
}
violation[{"msg": msg}] {
  input.review.namespaceObject.metadata.labels[k]
  msg := sprintf("namespace label %v", [k])
}
violation[{"msg": msg}] {
  not input.review.namespace
  msg := sprintf("no namespace in the request for %v", [input.review.kind.kind])
}
"""}]}}


def _review_shapes(n, seed):
    """every input shape of HandleReview (target.go:81-138) over synthetic objects: bare objects with Operation / Source, AdmissionRequests
    (CREATE / UPDATE / DELETE / CONNECT; object, oldObject, both, neither; envelope members present, null, missing), objects without a
    kind, DELETE without oldObject (HandleReview's error), the reviews.Namespace option"""
    from gatekeeper_amd import driver as D
    nss = synth.gen_namespaces()
    rng = synth.SplitMix64(seed)
    out = []
    for i, o in enumerate(synth.gen_objects(n, seed=seed, mixed=True)):
        ns = synth.namespace_for(o, nss)
        shape = rng.below(10)
        source = ["Original", "Generated", "All", ""][rng.below(8) % 4] if rng.below(4) == 0 else "Original"
        nsobj = ns if rng.below(5) == 0 else None
        if shape == 0:
            out.append(D.to_review_in(D.AugmentedUnstructured(D.Unstructured(o), ns, source, ["", "CREATE", "DELETE", "UPDATE"][rng.below(4)]), nsobj))
            continue
        if shape == 1:
            out.append(D.to_review_in(D.Unstructured(o), nsobj))
            continue
        api = o.get("apiVersion", "v1")
        g, _, ver = api.rpartition("/")
        req = {"uid": "u-%d" % i, "kind": {"group": g, "version": ver, "kind": o["kind"]}, "name": o["metadata"]["name"],
               "operation": ["CREATE", "UPDATE", "DELETE", "CONNECT"][rng.below(4)]}
        if rng.below(3):
            req["userInfo"] = {"username": "user-%d" % rng.below(5), "groups": ["system:authenticated"]}
        if o["metadata"].get("namespace") and rng.below(6):
            req["namespace"] = o["metadata"]["namespace"]
        kind_of = rng.below(8)
        if kind_of == 0:
            req["oldObject"] = o
        elif kind_of == 1:
            old = json.loads(json.dumps(o))
            old["metadata"]["labels"] = {"was": "old"}
            req["object"], req["oldObject"] = o, old
        elif kind_of == 2:
            req["object"], req["oldObject"] = None, None
        elif kind_of == 3:
            nokind = {k: v for k, v in o.items() if k != "kind"}
            req["object"] = nokind
        else:
            req["object"] = o
        if rng.below(3) == 0:
            req.update({"dryRun": bool(rng.below(2)), "requestKind": {"group": g, "version": ver, "kind": o["kind"]}, "options": {"kind": "CreateOptions"}})
        if shape == 2:
            out.append(D.to_review_in(D.AdmissionRequest(req), nsobj))
        else:
            out.append(D.to_review_in(D.AugmentedReview(D.AdmissionRequest(req), ns if rng.below(8) else None, source), nsobj))
    return out


def _c_reviews(rins):
    from gatekeeper_amd import _lib as L
    from gatekeeper_amd import driver as D
    arr = (L.gk_review_in * max(1, len(rins)))()
    for a, r in zip(arr, rins):
        D.Engine._fill(a, r)
    return arr


def test_review_shapes_checker_equals_the_python_oracle(fixtures):
    """HandleReview + Matcher.Match + input.review of the checker, pinned against the Python oracle on every input shape"""
    from gatekeeper_amd import driver as D
    from gatekeeper_amd import _lib as L
    ts = synth.psp_templates(fixtures) + [ENVELOPE_TEMPLATE]
    cs = synth.audit_constraints() + [{"apiVersion": "constraints.gatekeeper.sh/v1beta1", "kind": "K8sEnvelope", "metadata": {"name": "envelope"}, "spec": {}},
                                      {"apiVersion": "constraints.gatekeeper.sh/v1beta1", "kind": "K8sEnvelope", "metadata": {"name": "envelope-generated"},
                                       "spec": {"match": {"source": "Generated", "kinds": [{"apiGroups": [""], "kinds": ["Pod"]}]}}}]
    rins = _review_shapes(500, 71)
    oc = OC.Client()
    for t in ts:
        oc.add_template(t)
    for c in cs:
        oc.add_constraint(c)
    keys = {(c["kind"], c["metadata"]["name"]): i for i, c in enumerate(cs)}
    src_name = {v: k for k, v in D._SOURCES.items()}
    want_v, want_e, want_rej = set(), set(), set()
    for i, r in enumerate(rins):
        body, ns = json.loads(r.json), json.loads(r.namespace) if r.namespace else None
        nsobj = json.loads(r.ns_object) if r.ns_object else None
        source = src_name.get(r.source, "invalid")
        op = r.operation.decode() if r.operation else ""
        shape = OT.AugmentedUnstructured(OT.Unstructured(body), ns, source, op) if r.kind == L.GK_REVIEW_OBJECT else OT.AugmentedReview(OT.AdmissionRequest(body), ns, source)
        try:
            results = oc.review(shape, OC.AUDIT_EP, namespace=nsobj)
        except OT.ReviewError:
            want_rej.add(i)
            continue
        for res in results:
            row = keys[(res.constraint["kind"], res.constraint["metadata"]["name"])]
            (want_e if res.msg.startswith("unable to match constraints: ") and not res.metadata.get("details") else want_v).add((row, i))
    ck = IndepChecker(ts, cs)
    viol, err, rejected = ck.check_reviews(_c_reviews(rins), len(rins), threads=3)
    assert {int(i) for i in np.nonzero(rejected)[0]} == want_rej and len(want_rej) > 5
    got_v, got_e = _pairs_of(viol, len(rins)), _pairs_of(err, len(rins))
    assert len(want_v) > 500 and len(want_e) > 20
    assert got_v == want_v, (sorted(got_v - want_v)[:5], sorted(want_v - got_v)[:5])
    assert got_e == want_e, (sorted(got_e - want_e)[:5], sorted(want_e - got_e)[:5])


def test_review_shapes_product_equals_the_compiled_checker(fixtures):
    """20 000 reviews of every HandleReview shape x 52 constraints: the product's bitmaps and review statuses (CPU build of the engine)
    against the independent compiled checker"""
    from gatekeeper_amd import driver as D
    from gatekeeper_amd import _lib as L
    ts = synth.psp_templates(fixtures) + [ENVELOPE_TEMPLATE]
    cs = synth.audit_constraints() + [{"apiVersion": "constraints.gatekeeper.sh/v1beta1", "kind": "K8sEnvelope", "metadata": {"name": "envelope"}, "spec": {}},
                                      {"apiVersion": "constraints.gatekeeper.sh/v1beta1", "kind": "K8sEnvelope", "metadata": {"name": "envelope-generated"},
                                       "spec": {"match": {"source": "Generated", "kinds": [{"apiGroups": [""], "kinds": ["Pod"]}]}}}]
    n = 20000
    rins = _review_shapes(n, 72)
    drv = D.Driver(device=0, hostemu=True)
    client = D.Client(drv)
    for t in ts:
        client.AddTemplate(t)
    for c in cs:
        client.AddConstraint(c)
    table = drv.engine.create_table(rins, keep_docs=False)
    ev = table.eval()
    ids = [drv.constraint_id(client.constraints[(k["kind"], k["metadata"]["name"])]) for k in cs]
    ck = IndepChecker(ts, cs)
    viol, err, rejected = ck.check_reviews(_c_reviews(rins), n, threads=4)
    product_rejected = {i for i, st in enumerate(table.statuses) if st != L.GK_OK}
    assert product_rejected == {int(i) for i in np.nonzero(rejected)[0]} and len(product_rejected) > 100
    assert len(ev.too_big_reviews()) == 0
    row_of = {int(cid): i for i, cid in enumerate(ev.constraint_ids)}
    words = (n + 63) // 64
    tail = np.uint64((1 << (n % 64)) - 1) if n % 64 else None
    pairs = errs = 0
    for row, cid in enumerate(ids):
        d_v, d_e = np.array(ev.viol[row_of[cid]][:words], copy=True), np.array(ev.err[row_of[cid]][:words], copy=True)
        if tail is not None:
            d_v[-1] &= tail
            d_e[-1] &= tail
        assert (d_v == viol[row]).all(), ("viol", cs[row]["kind"], cs[row]["metadata"]["name"], [int(i) for i in np.nonzero(np.unpackbits((d_v ^ viol[row]).view(np.uint8), bitorder="little"))[0][:5]])
        assert (d_e == err[row]).all(), ("err", cs[row]["kind"], cs[row]["metadata"]["name"], [int(i) for i in np.nonzero(np.unpackbits((d_e ^ err[row]).view(np.uint8), bitorder="little"))[0][:5]])
        pairs += int(np.unpackbits(d_v.view(np.uint8)).sum())
        errs += int(np.unpackbits(d_e.view(np.uint8)).sum())
    assert pairs > n and errs > 1000
    table.free()


@pytest.mark.parametrize("policy,n", [("audit-50", 6000), ("corpus-200", 2500)])
def test_result_totals_product_equals_the_compiled_checker(policy, n, fixtures):
    """gk_table_totals (the audit's totalViolationsPerConstraint: RESULTS, several per violating pair -- counted on the device where the
    plan can, rendered on the host where it cannot) against the independent compiled checker's count of distinct (msg, details) per pair"""
    from gatekeeper_amd import driver as D
    ts, cs = (synth.psp_templates(fixtures), synth.audit_constraints()) if policy == "audit-50" else synth.corpus(fixtures)
    drv = D.Driver(device=0, hostemu=True)
    client = D.Client(drv)
    for t in ts:
        client.AddTemplate(t)
    for c in cs:
        client.AddConstraint(c)
    batch = synth.NativeBatch(drv.engine.lib, n, seed=synth.SEED + 21, mixed=True, start=0, namespaces=synth.gen_namespaces())
    table = drv.engine.create_table_native(batch.reviews, n, keep_docs=False, resident=True, keep_text=True, pruned=True)
    table.eval()
    product = table.totals()
    ids = [drv.constraint_id(client.constraints[(k["kind"], k["metadata"]["name"])]) for k in cs]
    ck = IndepChecker(ts, cs)
    viol, _err, results = ck.check_totals(batch.reviews, n, threads=4)
    for row, cid in enumerate(ids):
        pairs = int(np.unpackbits(viol[row].view(np.uint8)).sum())
        assert product[cid] == (int(results[row]), pairs), (cs[row]["kind"], cs[row]["metadata"]["name"], product[cid], int(results[row]), pairs)
    assert int(results.sum()) > sum(p for _, p in product.values()) > n     # several results per pair do occur
    assert table.rendered_pairs < sum(p for _, p in product.values()) // 5  # ... and most pairs were counted, not rendered
    table.free()
