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
    expected = []
    for o in objs[:-1]:
        res = oc.review(OT.AugmentedUnstructured(OT.Unstructured(o), synth.namespace_for(o, nss), "Original"), OC.WEBHOOK_EP)
        expected.append(sorted((r.constraint["metadata"]["name"], r.msg) for r in res))
    got, sizes, errors = [None] * len(objs), [0] * len(objs), []

    def worker(w):
        try:
            for i in range(w, len(objs), 16):
                rv = D.AugmentedUnstructured(D.Unstructured(objs[i]), synth.namespace_for(objs[i], nss), "Original")
                try:
                    resp = c.driver.Query(D.TARGET_NAME, cons, rv)
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
    assert isinstance(got[-1], D.LimitError)          # the review beyond the limits fails closed, its batch mates are answered
    assert got[:-1] == expected
    assert sum(len(e) for e in expected) > 0
    assert max(sizes) > 1                              # calls really shared launches
    # a review HandleReview rejects is an error for that caller only
    with pytest.raises(D.EngineError):
        c.driver.Query(D.TARGET_NAME, cons, D.AugmentedReview(D.AdmissionRequest({"operation": "DELETE", "object": objs[0]}), None, "Original"))
    assert c.driver.Query(D.TARGET_NAME, cons[:3], D.AugmentedUnstructured(D.Unstructured(objs[0]), None, "Original")).results is not None


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
