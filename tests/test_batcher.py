"""Row f1: the micro-batcher behind the C ABI (gk_query).  Driver.Query is one review at a time
(pkg/drivers/k8scel/driver.go:162-251) and the webhook calls it from many goroutines (pkg/webhook/policy.go:142-146);
here 16 threads call it concurrently, the engine coalesces them into shared launches, and every caller gets exactly the
results the oracle computes for its own review."""
import threading

import pytest

from gatekeeper_amd import driver as D
from gatekeeper_amd import synth
from oracle import client as OC
from oracle import target as OT
from parity_util import BACKENDS, key, load_both


@pytest.mark.parametrize("backend", [b for b in BACKENDS if b.id in ("hostemu", "gpu")])
def test_concurrent_queries_are_batched_and_exact(backend, fixtures):
    c, oc = load_both(backend, synth.psp_templates(fixtures), synth.psp_constraints())
    nss = synth.gen_namespaces()
    objs = synth.gen_objects(320, seed=61)
    objs.append({"apiVersion": "v1", "kind": "Pod", "metadata": {"name": "huge"}, "spec": {"containers": [{"name": "c%d" % i, "image": "x"} for i in range(300)]}})
    cons = list(c.constraints.values())
    c.driver.StartBatcher(max_batch=32, window_us=2000)
    objs[-1]["spec"]["containers"][41]["securityContext"] = {"privileged": True}   # (a violation hidden among 300 containers)
    expected = []
    for o in objs:
        res = oc.review(OT.AugmentedUnstructured(OT.Unstructured(o), synth.namespace_for(o, nss), "Original"), OC.WEBHOOK_EP)
        expected.append(sorted((r.constraint["metadata"]["name"], r.msg) for r in res))
    got, sizes, errors = [None] * len(objs), [0] * len(objs), []

    def worker(w):
        try:
            for i in range(w, len(objs), 16):
                rv = D.AugmentedUnstructured(D.Unstructured(objs[i]), synth.namespace_for(objs[i], nss), "Original")
                try:
                    resp = c.driver.QueryMatching(D.TARGET_NAME, cons, rv)
                    got[i] = sorted((r.constraint["metadata"]["name"], r.msg) for r in resp.results)
                except D.LimitError as e:
                    got[i] = e
                sizes[i] = c.driver.last_query_stats["batch_size"]
        except Exception as e:   # noqa: BLE001
            errors.append(e)

    threads = [threading.Thread(target=worker, args=(w,)) for w in range(16)]
    for t in threads:
        t.start()
    for t in threads:
        t.join()
    assert not errors, errors
    # the review beyond the DEVICE's limits (300 containers) is answered by the engine's host evaluator (round 5; refused until then),
    # its batch mates by the device as ever
    assert got == expected and expected[-1]
    assert sum(len(e) for e in expected) > 0
    assert max(sizes) > 1                              # calls really shared launches
    # a review HandleReview rejects is an error for that caller only
    with pytest.raises(D.EngineError):
        c.driver.QueryMatching(D.TARGET_NAME, cons, D.AugmentedReview(D.AdmissionRequest({"operation": "DELETE", "object": objs[0]}), None, "Original"))
    assert c.driver.QueryMatching(D.TARGET_NAME, cons[:3], D.AugmentedUnstructured(D.Unstructured(objs[0]), None, "Original")).results is not None


@pytest.mark.parametrize("requests", [False, True])
@pytest.mark.parametrize("backend", [b for b in BACKENDS if b.id in ("hostemu", "gpu")])
def test_native_query_storm_counts_match_the_oracle(backend, fixtures, requests):
    """The load generator of the f1 measurement (gk_synth_query_storm: native threads calling gk_query): every call
    succeeds, calls share launches, and the number of results returned equals what the oracle finds for the same reviews
    at the same enforcement point set (gk_query answers for every loaded constraint)."""
    c, oc = load_both(backend, synth.psp_templates(fixtures), synth.psp_constraints())
    nss = synth.gen_namespaces()
    n = 192
    batch = synth.NativeBatch(c.driver.engine.lib, n, seed=67, namespaces=nss, requests=requests)   # requests: the webhook's AdmissionRequest JSON
    want = 0
    for i, o in enumerate(synth.gen_objects(n, seed=67)):
        if requests:
            rv = OT.AugmentedReview(OT.AdmissionRequest(synth.admission_request_for(o, i)), synth.namespace_for(o, nss), "Original")
        else:
            rv = OT.AugmentedUnstructured(OT.Unstructured(o), synth.namespace_for(o, nss), "Original")
        want += len(oc.review(rv, OC.AUDIT_EP))
    c.driver.StartBatcher(max_batch=32, window_us=1000)
    out = batch.query_storm(c.driver.engine, threads=8, per_thread=n // 8)
    assert out["calls"] == n and out["errors"] == 0
    assert out["results"] == want and want > 50
    assert out["mean_batch"] > 1.5 and out["p50_us"] > 0 and out["p99_us"] >= out["p50_us"]


def test_query_stats_entries_and_their_descriptions(fixtures):
    """drivers.QueryResponse.StatsEntries + Driver.GetDescriptionForStat (pkg/drivers/k8scel/driver.go:231-247,257-264): one device
    launch answers every constraint of the review's batch, so the engine reports per-review figures (gk_query_stats) at template scope;
    every stat it names has a description, an unknown name is an error -- and the removal half of drivers.Driver (RemoveConstraint,
    RemoveTemplate, RemoveData) leaves an engine that answers like a fresh one"""
    c, oc = load_both("hostemu", synth.psp_templates(fixtures), synth.psp_constraints())
    nss = synth.gen_namespaces()
    pod = next(o for o in synth.gen_objects(50, seed=9) if o["spec"].get("hostNetwork") or any((x.get("securityContext") or {}).get("privileged") for x in o["spec"]["containers"]))
    rv = D.AugmentedUnstructured(D.Unstructured(pod), synth.namespace_for(pod, nss), "Original")
    cons = list(c.constraints.values())
    resp = c.driver.QueryMatching(D.TARGET_NAME, cons, rv, stats_enabled=True)
    # one entry per template kind of the queried constraints, in the Rego driver's shape (pkg/gator/test/test_test.go:357-391)
    kinds = sorted({k["kind"] for k in cons})
    assert len(resp.results) > 0 and [e["statsFor"] for e in resp.stats_entries] == kinds and not c.driver.QueryMatching(D.TARGET_NAME, cons, rv).stats_entries
    entry = resp.stats_entries[0]
    names = [s["name"] for s in entry["stats"]]
    assert entry["scope"] == "template" and names == [D.Driver.RUN_TIME_NS, D.Driver.CONSTRAINT_COUNT]
    assert all(isinstance(s["value"], int) and s["value"] >= 1 and s["source"] == {"type": "engine", "value": "Rego"} for s in entry["stats"])
    assert sum(e["stats"][1]["value"] for e in resp.stats_entries) == len(cons)
    assert entry["labels"] == [{"name": "TracingEnabled", "value": False}, {"name": "PrintEnabled", "value": False}, {"name": "target", "value": D.TARGET_NAME}]
    for n in names:
        assert c.driver.GetDescriptionForStat(n)
    with pytest.raises(D.ClientError, match="unknown stat name"):
        c.driver.GetDescriptionForStat("nope")
    # removal: one constraint, then a whole template (its constraints go with it), then a synced Namespace
    before = sorted(key(r) for r in c.Review(rv, D.WEBHOOK_EP))
    gone = cons[0]
    c.RemoveConstraint(gone)
    oc.remove_constraint(gone)
    want = sorted(key(r) for r in oc.review(OT.AugmentedUnstructured(OT.Unstructured(pod), synth.namespace_for(pod, nss), "Original"), OC.WEBHOOK_EP))
    assert sorted(key(r) for r in c.Review(rv, D.WEBHOOK_EP)) == want
    tmpl = next(t for t in synth.psp_templates(fixtures) if t["spec"]["crd"]["spec"]["names"]["kind"] == "K8sPSPPrivilegedContainer")
    c.RemoveTemplate(tmpl)
    oc.remove_template(tmpl)
    want = sorted(key(r) for r in oc.review(OT.AugmentedUnstructured(OT.Unstructured(pod), synth.namespace_for(pod, nss), "Original"), OC.WEBHOOK_EP))
    got = sorted(key(r) for r in c.Review(rv, D.WEBHOOK_EP))
    assert got == want and len(got) <= len(before)
    ns = nss[pod["metadata"]["namespace"]]
    c.AddData(ns)
    c.RemoveData(ns)
    assert sorted(key(r) for r in c.Review(rv, D.WEBHOOK_EP)) == want
    assert "constraints=" in c.driver.Dump()
