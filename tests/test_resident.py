"""Row f2: the resident-set audit.  Objects reach the driver through AddData (pkg/cachemanager/cachemanager.go:310-343);
pkg/audit's auditFromCache then reviews each of them serially (pkg/audit/manager.go:591-642).  Here the engine keeps them
flattened in HBM, one sweep evaluates all constraints over the set, and every object's results are a column of the sweep's
bitmaps.  Checked against the oracle's serial loop: the same objects, the same Namespace lookup, bit for bit -- before and
after 1 % of the objects are changed, some removed, some added and a Namespace relabelled."""
import json

import numpy as np
import pytest

from gatekeeper_amd import _lib as L
from gatekeeper_amd import driver as D
from gatekeeper_amd import synth
from oracle import client as OC
from oracle import cpu_ref as CR
from oracle import target as OT
from parity_util import BACKENDS, key, load_both


def _oracle_serial_loop(oc, cached):
    """auditFromCache: nsMap from the cached Namespaces, one Review per object with AugmentedUnstructured{obj, ns}"""
    ns_map = {o["metadata"]["name"]: o for o in cached.values() if o.get("kind") == "Namespace" and o.get("apiVersion") == "v1"}
    out = {}
    for path, o in cached.items():
        ns = ns_map.get((o.get("metadata") or {}).get("namespace") or "")
        out[path] = oc.review(OT.AugmentedUnstructured(OT.Unstructured(o), ns, ""), OC.AUDIT_EP, ns)
    return out


def _compare(c, oc):
    got, sweep = c.AuditFromCache()
    want = _oracle_serial_loop(oc, c.cached)
    assert set(got) == set(want)
    n = 0
    for path in want:
        a, b = sorted(key(r) for r in got[path]), sorted(key(r) for r in want[path])
        assert a == b, (path, a, b)
        n += len(b)
    return n, sweep


@pytest.mark.parametrize("backend", [b for b in BACKENDS if b.id in ("hostemu", "gpu")])
def test_audit_from_cache_incremental(backend, fixtures):
    c, oc = load_both(backend, synth.psp_templates(fixtures), synth.audit_constraints())
    nss = synth.gen_namespaces()
    objs = [o for o in synth.gen_objects(700, seed=31, mixed=True)]
    for o in list(nss.values()) + objs:
        c.AddData(o)
        oc.add_data(o)
    n0, s0 = _compare(c, oc)
    assert n0 > 100 and s0["flattened"] == len(c.cached) and s0["n_chunks"] == 1
    # nothing changed: the second sweep flattens nothing and re-evaluates nothing
    n1, s1 = _compare(c, oc)
    assert n1 == n0 and s1["flattened"] == 0
    # 1 % of the objects change, two disappear, three appear, one Namespace is relabelled
    rng = synth.SplitMix64(5)
    pods = [o for o in objs if o["kind"] == "Pod"]
    for k in range(7):
        o = json.loads(json.dumps(pods[rng.below(len(pods))]))
        o["spec"]["hostNetwork"] = True
        o["spec"]["containers"][0].setdefault("securityContext", {})["privileged"] = k % 2 == 0
        c.AddData(o); oc.add_data(o)
    for o in (pods[3], pods[11]):
        c.RemoveData(o); oc.remove_data(o)
    for o in synth.gen_objects(3, seed=99, start=5000):
        c.AddData(o); oc.add_data(o)
    relabel = json.loads(json.dumps(nss["prod-03"]))
    relabel["metadata"]["labels"]["env"] = "dev"
    relabel["metadata"]["labels"]["pci"] = "true"
    c.AddData(relabel); oc.add_data(relabel)
    n2, s2 = _compare(c, oc)
    in_ns = sum(1 for o in c.cached.values() if (o.get("metadata") or {}).get("namespace") == "prod-03")
    assert 0 < s2["flattened"] <= 7 + 3 + 1 + in_ns and s2["n_chunks"] == 2 and n2 != n0
    # a policy change invalidates the cached answers (and the next sweep re-evaluates without re-flattening)
    k_ = synth.audit_constraints()[0]
    c.RemoveConstraint(k_); oc.remove_constraint(k_)
    some = next(iter(c.cached))
    assert c.driver.ResidentReview(list(some)) is None
    n3, s3 = _compare(c, oc)
    assert s3["flattened"] == 0 and n3 <= n2
    # Driver.Query for a review that IS a swept resident object answers from the bitmap column: no table, no launch
    path = next(p for p, o in c.cached.items() if o.get("kind") == "Pod" and (o["metadata"].get("namespace") in nss))
    o = c.cached[path]
    ns = relabel if o["metadata"]["namespace"] == "prod-03" else nss[o["metadata"]["namespace"]]
    resp = c.driver.QueryMatching(D.TARGET_NAME, list(c.constraints.values()), D.AugmentedUnstructured(D.Unstructured(o), ns, ""), ns)
    assert c.driver.last_query_stats["batch_size"] == 0          # served from the resident set
    want = oc.review(OT.AugmentedUnstructured(OT.Unstructured(o), ns, ""), OC.AUDIT_EP, ns)
    assert sorted((r.constraint["metadata"]["name"], r.msg) for r in resp.results) == sorted((r.constraint["metadata"]["name"], r.msg) for r in want if not r.msg.startswith("unable to match"))


@pytest.mark.parametrize("backend", [b for b in BACKENDS if b.id in ("hostemu", "gpu")])
def test_resident_set_10k_against_compiled_reference(backend, fixtures):
    """10 000 synced objects, 1 % mutated: per-constraint violating-pair and RESULT totals of the sweep == the independent compiled
    checker (oracle/libgkindep.so) and the compiled loop around the product's evaluator (oracle/cpu_ref.cpp) over the same objects."""
    c, oc = load_both(backend, synth.psp_templates(fixtures), synth.audit_constraints())
    nss = synth.gen_namespaces()
    objs = synth.gen_objects(10000, seed=77, mixed=True)
    for o in list(nss.values()) + objs:
        c.AddData(o)
    s0 = c.driver.ResidentSweep()
    rng = synth.SplitMix64(11)
    for _ in range(100):
        i = rng.below(len(objs))
        o = json.loads(json.dumps(objs[i]))
        if "spec" in o and isinstance(o["spec"], dict):
            o["spec"]["hostPID"] = True
        o.setdefault("metadata", {}).setdefault("labels", {})["canary"] = "x"
        objs[i] = o
        c.AddData(o)
    s1 = c.driver.ResidentSweep(result_totals=True)
    assert s0["flattened"] == 10100 and 0 < s1["flattened"] <= 100 and s1["n_objects"] == 10100
    # the same reviews through the compiled reference loop
    cons = [c.constraints[(k["kind"], k["metadata"]["name"])] for k in synth.audit_constraints()]
    ref = CR.CpuRef(synth.psp_templates(fixtures), cons)
    all_objs = list(nss.values()) + objs
    arr = (L.gk_review_in * len(all_objs))()
    keep = []
    for i, o in enumerate(all_objs):
        body = json.dumps(o).encode()
        keep.append(body)
        arr[i].kind, arr[i].source, arr[i].json, arr[i].json_len = L.GK_REVIEW_OBJECT, L.GK_SRC_EMPTY, body, len(body)
        nsn = (o.get("metadata") or {}).get("namespace")
        if nsn in nss:
            nsb = json.dumps(nss[nsn]).encode()
            keep.append(nsb)
            arr[i].namespace_json, arr[i].namespace_len, arr[i].ns_object_json, arr[i].ns_object_len = nsb, len(nsb), nsb, len(nsb)
    out = ref.review(arr, len(all_objs), 4)
    pairs = {c.driver.constraint_id(k): int(np.unpackbits(out["viol"][row].view(np.uint8)).sum()) for row, k in enumerate(cons)}
    results = {c.driver.constraint_id(k): int(out["results"][row]) for row, k in enumerate(cons)}
    assert s1["pairs"] == pairs and s1["results"] == results and sum(pairs.values()) > 1000
    # ... and through the INDEPENDENT compiled checker (oracle/indep_check.cpp: nothing of the product linked; the loop above shares the
    # product's JSON reader and Rego evaluator): the same pair and RESULT totals per constraint
    from oracle.indep_check import IndepChecker
    ck = IndepChecker(synth.psp_templates(fixtures), synth.audit_constraints())
    viol, err, res = ck.check_totals(arr, len(all_objs), 4)
    ck.close()
    ipairs = {c.driver.constraint_id(k): int(np.unpackbits(viol[row].view(np.uint8)).sum()) for row, k in enumerate(cons)}
    iresults = {c.driver.constraint_id(k): int(res[row]) for row, k in enumerate(cons)}
    assert s1["pairs"] == ipairs and s1["results"] == iresults
