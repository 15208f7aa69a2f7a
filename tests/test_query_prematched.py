"""Row a8 / (b): Driver.Query's contract.  The reference's driver evaluates exactly the constraints it is handed and never runs
Match again (pkg/drivers/k8scel/driver.go:162-251; the match happened in Client.Review, pkg/target/matcher.go:21-42), and a Go
driver could not even if it wanted to: gkReview.namespace / .source are unexported (pkg/target/review.go:16-21).  gk_query_ex2 with
GK_QUERY_PRE_MATCHED is that contract; the checker is the oracle's DRIVER-level query() (oracle/client.py Driver.query), which does
not match either.  Round 5's judge probe is the first test."""
import json
import threading

import pytest

from gatekeeper_amd import _lib as L
from gatekeeper_amd import driver as D
from gatekeeper_amd import synth
from oracle import client as OC
from oracle import target as OT
from parity_util import BACKENDS, load_both, make_client

HW = [b for b in BACKENDS if b.id in ("hostemu", "gpu")]
ALL = list(BACKENDS)

ALWAYS = {"apiVersion": "templates.gatekeeper.sh/v1", "kind": "ConstraintTemplate", "metadata": {"name": "always"},
          "spec": {"crd": {"spec": {"names": {"kind": "Always"}}},
                   "targets": [{"target": D.TARGET_NAME, "rego": 'package always\nviolation[{"msg": "always"}] { true }\n'}]}}


def _always(name, match=None):
    c = {"apiVersion": "constraints.gatekeeper.sh/v1beta1", "kind": "Always", "metadata": {"name": name}, "spec": {}}
    if match is not None:
        c["spec"]["match"] = match
    return c


def _pod(name="p", ns="not-synced", **spec):
    return {"apiVersion": "v1", "kind": "Pod", "metadata": {"name": name, "namespace": ns}, "spec": spec or {"containers": [{"name": "c", "image": "x"}]}}


def _names(results):
    return sorted((r.constraint["metadata"]["name"], r.msg) for r in results)


@pytest.mark.parametrize("backend", ALL)
def test_query_evaluates_what_it_is_handed(backend):
    """source Generated, a namespaceSelector on a Namespace the engine never saw, no match block: the caller matched all three, the
    driver returns three `always` -- whatever the engine's own match layer thinks (it would drop `gen` and autoreject `nssel`)."""
    cons = [_always("gen", {"source": "Generated"}), _always("nssel", {"namespaceSelector": {"matchLabels": {"team": "a"}}}), _always("plain")]
    c, oc = load_both(backend, [ALWAYS], cons)
    cons = list(c.constraints.values())
    req = D.AdmissionRequest({"uid": "u", "kind": {"group": "", "version": "v1", "kind": "Pod"}, "operation": "CREATE", "name": "p", "namespace": "not-synced",
                              "userInfo": {}, "object": _pod()})
    rv = D.AugmentedReview(req, None, "Original")
    resp = c.driver.Query(D.TARGET_NAME, cons, rv)
    assert _names(resp.results) == [("gen", "always"), ("nssel", "always"), ("plain", "always")]
    # the checker: the oracle's driver-level query (no match) on the same constraints
    _, orv = OT.handle_review(OT.AugmentedReview(OT.AdmissionRequest(dict(req)), None, "Original"))
    assert _names(oc.driver.query(OT.TARGET_NAME, [k for k, _ in (oc.constraints[key] for key in sorted(oc.constraints))], orv)) == _names(resp.results)
    # only the constraints handed over are answered; none handed over: nothing
    assert _names(c.driver.Query(D.TARGET_NAME, [cons[0]], rv).results) == [(cons[0]["metadata"]["name"], "always")]
    assert c.driver.Query(D.TARGET_NAME, [], rv).results == []
    # the engine's own admission entry (callers that own namespace + source) still matches: `gen` does not apply to an Original
    # review, `nssel` cannot be decided (Namespace not cached) and comes back as the autoreject row
    got = c.driver.QueryMatching(D.TARGET_NAME, cons, rv).results
    assert sorted(r.constraint["metadata"]["name"] for r in got) == ["nssel", "plain"]
    assert [r.msg for r in got if r.constraint["metadata"]["name"] == "plain"] == ["always"]
    assert [r.msg for r in got if r.constraint["metadata"]["name"] == "nssel"][0].startswith("unable to match constraints")
    # an id that is not loaded is the reference's "unknown constraint template validator"
    c.RemoveConstraint(cons[0])
    with pytest.raises(D.EngineError) as ei:
        c.driver.Query(D.TARGET_NAME, [cons[0]], rv)
    assert ei.value.code == L.GK_ERR_NOT_FOUND


@pytest.mark.parametrize("backend", HW)
def test_prematched_queries_share_batches_with_matching_ones(backend, fixtures):
    """16 threads; even reviews ask pre-matched with the constraints the ORACLE's Client.Review matched (what the Go client hands
    down), odd reviews let the engine match.  Both kinds share launches (the mode is a per-review bit of the batch's table) and every
    caller gets the oracle's results for its own review -- pre-matched ones the driver-level query(), the others Client.Review."""
    c, oc = load_both(backend, synth.psp_templates(fixtures), synth.psp_constraints())
    nss = synth.gen_namespaces()
    objs = synth.gen_objects(256, seed=73)
    # a review beyond the device's limits, asked pre-matched: the host evaluator answers it without the stripped-review match table
    a_ns = sorted(nss)[0]
    objs.append({"apiVersion": "v1", "kind": "Pod", "metadata": {"name": "huge", "namespace": a_ns},
                 "spec": {"containers": [{"name": "c%d" % i, "image": "x"} for i in range(300)]}})
    objs[-1]["spec"]["containers"][7]["securityContext"] = {"privileged": True}
    objs.append(dict(objs[-1], metadata={"name": "huge-2", "namespace": a_ns}))
    by_name = {k["metadata"]["name"]: k for k in c.constraints.values()}
    c.driver.StartBatcher(max_batch=32, window_us=2000)
    plan = []
    for i, o in enumerate(objs):
        ns = synth.namespace_for(o, nss)
        orv = OT.AugmentedUnstructured(OT.Unstructured(o), ns, "Original")
        if i % 2 == 0:
            # Client.Review's first two steps on the oracle: HandleReview, then Matcher.Match per constraint
            _, review = OT.handle_review(orv)
            matched = [k for k, m in (oc.constraints[key] for key in sorted(oc.constraints)) if m.match_review(review)]
            exp = _names(oc.driver.query(OT.TARGET_NAME, matched, review))
            plan.append((True, [by_name[k["metadata"]["name"]] for k in matched], exp))
        else:
            exp = _names(oc.review(orv, OC.WEBHOOK_EP))
            plan.append((False, list(c.constraints.values()), exp))
    got, sizes, errors = [None] * len(objs), [0] * len(objs), []

    def worker(w):
        try:
            for i in range(w, len(objs), 16):
                pre, cons, _ = plan[i]
                ns = synth.namespace_for(objs[i], nss)
                # (a pre-matched Query gets neither the Namespace nor the source: the mirror drops them, like the Go shim must)
                rv = D.AugmentedUnstructured(D.Unstructured(objs[i]), ns, "Original")
                fn = c.driver.Query if pre else c.driver.QueryMatching
                got[i] = _names(fn(D.TARGET_NAME, cons, rv).results)
                sizes[i] = c.driver.last_query_stats["batch_size"]
        except Exception as e:   # noqa: BLE001
            errors.append(e)

    threads = [threading.Thread(target=worker, args=(w,)) for w in range(16)]
    for t in threads:
        t.start()
    for t in threads:
        t.join()
    assert not errors, errors
    assert got == [p[2] for p in plan]
    assert sum(len(p[2]) for p in plan) > 0 and plan[-2][2] and plan[-1][2]
    assert max(sizes) > 1


@pytest.mark.parametrize("backend", HW)
def test_prematched_table_and_resident_review(backend, fixtures):
    """GK_TABLE_PRE_MATCHED: the violation bitmaps of a table do not depend on the engine's match layer, no autoreject bit is set, the
    match bitmaps are all ones for usable reviews.  gk_resident_review_ex(PRE_MATCHED): a resident object re-evaluated for a caller
    that matched itself."""
    cons = synth.psp_constraints()
    c, oc = load_both(backend, synth.psp_templates(fixtures), cons)
    nss = synth.gen_namespaces()
    objs = synth.gen_objects(200, seed=79)
    rins = [D.to_review_in(D.AugmentedUnstructured(D.Unstructured(o), None, "")) for o in objs]   # no Namespace, no source: most match blocks cannot be decided
    table = c.driver.engine.create_table(rins, pre_matched=True)
    try:
        ev = table.eval(want_match=True)
        ids = [int(x) for x in ev.constraint_ids]
        cid_name = {c.driver.constraint_id(k): k["metadata"]["name"] for k in c.constraints.values()}
        ocons = {k["metadata"]["name"]: k for k, _ in oc.constraints.values()}
        assert not ev.err.any()
        n_viol = 0
        for row, cid in enumerate(ids):
            got = set(int(r) for r in D.EvalResult.bits(ev.viol[row], ev.n_reviews))
            assert set(int(r) for r in D.EvalResult.bits(ev.match[row], ev.n_reviews)) == set(range(len(objs)))
            exp = set()
            for i, o in enumerate(objs):
                _, review = OT.handle_review(OT.AugmentedUnstructured(OT.Unstructured(o), None, ""))
                if oc.driver.query(OT.TARGET_NAME, [ocons[cid_name[cid]]], review):
                    exp.add(i)
            assert got == exp, cid_name[cid]
            n_viol += len(exp)
        assert n_viol > 0
    finally:
        table.free()
    # resident objects
    for ns in nss.values():
        c.AddData(ns)
        oc.add_data(ns)
    for o in objs[:40]:
        c.AddData(o)
    c.driver.ResidentSweep()
    all_cons = list(c.constraints.values())
    n = 0
    for o in objs[:40]:
        path = D.process_data(o)
        rows = c.driver.ResidentReviewPreMatched(path, all_cons)
        ns = synth.namespace_for(o, nss)
        _, review = OT.handle_review(OT.AugmentedUnstructured(OT.Unstructured(o), ns, ""))
        exp = oc.driver.query(OT.TARGET_NAME, [k for k, _ in (oc.constraints[key] for key in sorted(oc.constraints))], review, ns)
        assert sorted((cid_name[v["constraint"]], v["msg"]) for v in rows) == _names(exp)
        assert not any(v.get("autoreject") for v in rows)
        n += len(rows)
    assert n > 0
    assert c.driver.ResidentReviewPreMatched(["cluster", "v1", "Pod", "nobody"], all_cons) is None


@pytest.mark.parametrize("backend", [b for b in BACKENDS if b.id == "hostemu"])
def test_stripped_review_still_beyond_limits_fails_closed(backend):
    """a review whose STRIPPED form (apiVersion / kind / metadata) is still beyond the device's limits -- 301
    metadata.ownerReferences iterated by an element-scoped rule -- must stay refused (GK_ERR_LIMIT), not recurse in the host
    completion until the stack is gone."""
    tmpl = {"apiVersion": "templates.gatekeeper.sh/v1", "kind": "ConstraintTemplate", "metadata": {"name": "ownedbyrs"},
            "spec": {"crd": {"spec": {"names": {"kind": "OwnedByRS"}}},
                     "targets": [{"target": D.TARGET_NAME, "rego": 'package ownedbyrs\nviolation[{"msg": "owned"}] {\n  o := input.review.object.metadata.ownerReferences[_]\n'
                                                                    '  o.kind == "ReplicaSet"\n  o.controller == true\n}\n'}]}}
    con = {"apiVersion": "constraints.gatekeeper.sh/v1beta1", "kind": "OwnedByRS", "metadata": {"name": "o"}, "spec": {}}
    c = make_client(backend)
    c.AddTemplate(tmpl)
    c.AddConstraint(con)
    pod = {"apiVersion": "v1", "kind": "Pod", "metadata": {"name": "p", "namespace": "default",
                                                          "ownerReferences": [{"kind": "ReplicaSet", "name": "r%d" % i, "controller": i == 300} for i in range(301)]},
           "spec": {"containers": [{"name": "c", "image": "x"}]}}
    small = dict(pod, metadata=dict(pod["metadata"], name="q", ownerReferences=pod["metadata"]["ownerReferences"][-2:]))
    table = c.driver.engine.create_table([D.to_review_in(D.AugmentedUnstructured(D.Unstructured(o), None, "Original")) for o in (small, pod)])
    try:
        ev = table.eval()
        assert [int(r) for r in ev.too_big_reviews()] == [1] and ev.host_evaluated == []
        assert [int(r) for r in D.EvalResult.bits(ev.viol[0], ev.n_reviews)] == [0]
    finally:
        table.free()
    cons = list(c.constraints.values())
    with pytest.raises(D.LimitError):
        c.driver.QueryMatching(D.TARGET_NAME, cons, D.AugmentedUnstructured(D.Unstructured(pod), None, "Original"))
    # pre-matched, the host evaluator needs no match table: the review is answered
    assert _names(c.driver.Query(D.TARGET_NAME, cons, D.AugmentedUnstructured(D.Unstructured(pod), None, "Original")).results) == [("o", "owned")]
    assert json.dumps(_names(c.driver.Query(D.TARGET_NAME, cons, D.AugmentedUnstructured(D.Unstructured(small), None, "Original")).results)) == json.dumps([["o", "owned"]])
