"""Row f3: the process excluder as a pre-filter of the batched paths.  pkg/controller/config/process/excluder.go (Add /
Replace / IsNamespaceExcluded), applied by pkg/audit/manager.go:599 (audit loop), :530 (listing) and
pkg/webhook/policy.go:197 + common.go:149-189 (validating webhook).  The oracle (oracle/excluder.py) is pinned on the
reference's own rows (excluder_test.go); the product is then compared with the oracle on the synthetic object stream."""
import json

import pytest

from gatekeeper_amd import driver as D
from gatekeeper_amd import synth
from oracle import client as OC
from oracle import excluder as OX
from oracle import target as OT
from parity_util import BACKENDS, key, load_both, make_client

HOSTEMU_GPU = [b for b in BACKENDS if b.id in ("hostemu", "gpu")]

# pkg/controller/config/process/excluder_test.go:12-68 (TestExactOrWildcardMatch)
WILDCARD_ROWS = [
    ("exact text match", ["kube-system", "foobar"], "kube-system", True),
    ("wildcard prefix match", ["kube-*", "foobar"], "kube-system", True),
    ("wildcard suffix match", ["*-system", "foobar"], "kube-system", True),
    ("lack of asterisk prevents globbing", ["kube-"], "kube-system", False),
]
# excluder_test.go:70-180 (TestGetExcludedNamespaces)
ADD_ROWS = [
    ("single process with multiple namespaces", [{"excludedNamespaces": ["kube-system", "kube-public"], "processes": ["audit"]}], "audit", ["kube-system", "kube-public"]),
    ("wildcard process affects all processes", [{"excludedNamespaces": ["kube-*", "default"], "processes": ["*"]}], "webhook", ["kube-*", "default"]),
    ("multiple match entries for same process", [{"excludedNamespaces": ["kube-system"], "processes": ["sync"]},
                                                 {"excludedNamespaces": ["monitoring"], "processes": ["sync"]}], "sync", ["kube-system", "monitoring"]),
    ("empty for non-configured process", [{"excludedNamespaces": ["kube-system"], "processes": ["audit"]}], "mutation-webhook", []),
    ("mixed processes with overlapping namespaces", [{"excludedNamespaces": ["kube-system", "app-*"], "processes": ["webhook", "mutation-webhook"]},
                                                     {"excludedNamespaces": ["monitoring"], "processes": ["webhook"]}], "webhook", ["kube-system", "app-*", "monitoring"]),
    ("empty excluder returns empty list", [], "audit", []),
]


def _pod(ns, name="p"):
    return {"apiVersion": "v1", "kind": "Pod", "metadata": {"name": name, "namespace": ns}}


def test_oracle_on_reference_rows():
    for name, pats, ns, want in WILDCARD_ROWS:
        ex = OX.Excluder([{"excludedNamespaces": pats, "processes": ["audit"]}])
        assert ex.match("audit", ns) is want, name
    for name, entries, proc, want in ADD_ROWS:
        assert OX.Excluder(entries).get_excluded_namespaces(proc) == sorted(want), name


def test_engine_on_reference_rows():
    c = make_client("hostemu")
    for name, pats, ns, want in WILDCARD_ROWS:
        c.SetExcluder([{"excludedNamespaces": pats, "processes": ["audit"]}])
        assert c.IsNamespaceExcluded("audit", D.Unstructured(_pod(ns))) is want, name
        assert c.IsNamespaceExcluded("webhook", D.Unstructured(_pod(ns))) is False, name
    probes = ["kube-system", "kube-public", "kube-x", "default", "monitoring", "app-1", "app", "other", ""]
    for name, entries, proc, want in ADD_ROWS:
        c.SetExcluder(entries)
        ox = OX.Excluder(entries)
        for p in OX.ALL_PROCESSES:
            for ns in probes:
                assert c.IsNamespaceExcluded(p, D.Unstructured(_pod(ns))) is ox.match(p, ns), (name, p, ns)
    # a core Namespace object is matched on its NAME, a Namespace of another group on its (empty) namespace (excluder.go:100-104)
    c.SetExcluder([{"excludedNamespaces": ["kube-*"], "processes": ["*"]}])
    assert c.IsNamespaceExcluded("audit", D.Unstructured({"apiVersion": "v1", "kind": "Namespace", "metadata": {"name": "kube-system"}}))
    assert not c.IsNamespaceExcluded("audit", D.Unstructured({"apiVersion": "x.io/v1", "kind": "Namespace", "metadata": {"name": "kube-system"}}))
    c.SetExcluder(None)
    assert not c.IsNamespaceExcluded("audit", D.Unstructured(_pod("kube-system")))


def test_webhook_view_of_admission_requests():
    """common.go:149-189: oldObject on DELETE, the request's namespace wins over the object's"""
    entries = [{"excludedNamespaces": ["kube-*", "*-skip"], "processes": ["webhook"]}]
    ox = OX.Excluder(entries)
    c = make_client("hostemu")
    c.SetExcluder(entries)
    reqs = [
        {"operation": "CREATE", "namespace": "kube-system", "object": _pod("default")},
        {"operation": "CREATE", "namespace": "default", "object": _pod("kube-system")},
        {"operation": "DELETE", "namespace": "kube-system", "oldObject": _pod("kube-system"), "object": None},
        {"operation": "DELETE", "namespace": "kube-system", "object": _pod("kube-system")},                      # oldObject missing: error -> reviewed
        {"operation": "UPDATE", "namespace": "", "object": {"apiVersion": "v1", "kind": "Namespace", "metadata": {"name": "to-skip"}}},
        {"operation": "CREATE", "namespace": "to-skip", "object": {"apiVersion": "v1", "kind": "Namespace", "metadata": {"name": "keep"}}},
        {"operation": "CREATE", "namespace": "kube-system", "object": {"metadata": {"name": "nokind"}}},          # does not decode
        {"operation": "CREATE", "namespace": "a-skip", "object": {"apiVersion": "apps/v1", "kind": "Deployment", "metadata": {"name": "d"}}},
    ]
    want = [ox.webhook_skips("webhook", r) for r in reqs]
    assert want == [True, False, True, False, True, False, False, True]
    for r, w in zip(reqs, want):
        full = dict({"uid": "u", "kind": {"group": "", "version": "v1", "kind": "Pod"}, "name": "p", "userInfo": {}}, **r)
        assert c.IsNamespaceExcluded("webhook", D.AdmissionRequest(full)) is w, r
        assert c.IsNamespaceExcluded("audit", D.AdmissionRequest(full)) is False


ENTRIES = [{"excludedNamespaces": ["kube-*", "prod-0*", "gen-ns-00000*"], "processes": ["audit"]},
           {"excludedNamespaces": ["dev-*"], "processes": ["webhook"]},
           {"excludedNamespaces": ["*-09", "team-1*"], "processes": ["*"]}]


def _oracle_results(oc, ox, process, rv):
    out = []
    for r in rv:
        if process and ox.is_namespace_excluded(process, r.object):
            out.append([])
        else:
            out.append(oc.review(OT.AugmentedUnstructured(OT.Unstructured(r.object), r.namespace, r.source), OC.AUDIT_EP, None))
    return out


@pytest.mark.parametrize("backend", HOSTEMU_GPU)
def test_batched_review_and_audit_honour_the_excluder(backend, fixtures):
    c, oc = load_both(backend, synth.psp_templates(fixtures), synth.audit_constraints())
    ox = OX.Excluder(ENTRIES)
    c.SetExcluder(ENTRIES)
    nss = synth.gen_namespaces()
    objs = synth.gen_objects(500, seed=23, mixed=True)
    rv = [D.AugmentedUnstructured(D.Unstructured(o), synth.namespace_for(o, nss), "Original") for o in objs]
    n_excl = {}
    for process in (None, "audit", "webhook"):
        got = c.ReviewBatch(rv, D.AUDIT_EP, None, process=process)
        want = _oracle_results(oc, ox, process, rv)
        n_excl[process] = sum(1 for r in rv if process and ox.is_namespace_excluded(process, r.object))
        for i, (g, w) in enumerate(zip(got, want)):
            assert sorted(key(r) for r in g) == sorted(key(r) for r in w), (process, i)
    assert n_excl["audit"] > 50 and n_excl["webhook"] > 50 and n_excl["audit"] != n_excl["webhook"]
    # the audit aggregation: totals over the objects the audit process keeps (manager.go:599-608, 885-941)
    want = _oracle_results(oc, ox, "audit", rv)
    totals, per_action = {}, {}
    for w in want:
        for r in w:
            k = (r.constraint["kind"], r.constraint["metadata"]["name"])
            totals[k] = totals.get(k, 0) + 1
            per_action[r.enforcement_action] = per_action.get(r.enforcement_action, 0) + 1
    rep = c.AuditAggregate(rv)
    assert not rep.errors
    got = {(k[0], k[2]): v["total"] for k, v in rep.items() if v["total"]}
    assert got == totals and rep.totals_per_action == per_action and len(per_action) >= 2
    c.SetExcluder([])
    rep2 = c.AuditAggregate(rv)
    assert sum(v["total"] for v in rep2.values()) > sum(totals.values())


@pytest.mark.parametrize("backend", HOSTEMU_GPU)
def test_resident_audit_honours_config_changes(backend, fixtures):
    """auditFromCache (manager.go:591-642): excluded objects are skipped; replacing the Config's match entries changes the
    answer of the next sweep without the objects being synced again"""
    c, oc = load_both(backend, synth.psp_templates(fixtures), synth.audit_constraints())
    nss = synth.gen_namespaces()
    objs = synth.gen_objects(400, seed=29, mixed=True)
    for o in list(nss.values()) + objs:
        c.AddData(o)
        oc.add_data(o)

    def compare(entries):
        ox = OX.Excluder(entries)
        c.SetExcluder(entries)
        got, sweep = c.AuditFromCache()
        ns_map = {o["metadata"]["name"]: o for o in c.cached.values() if o.get("kind") == "Namespace" and o.get("apiVersion") == "v1"}
        n = skipped = 0
        for path, o in c.cached.items():
            if ox.is_namespace_excluded("audit", o):
                want = []
                skipped += 1
            else:
                ns = ns_map.get((o.get("metadata") or {}).get("namespace") or "")
                want = oc.review(OT.AugmentedUnstructured(OT.Unstructured(o), ns, ""), OC.AUDIT_EP, ns)
            assert sorted(key(r) for r in got[path]) == sorted(key(r) for r in want), path
            n += len(want)
        assert int(sum(sweep["pairs"].values()) if isinstance(sweep.get("pairs"), dict) else 0) >= 0
        return n, skipped, sweep

    n_all, s_all, _ = compare([])
    n_ex, s_ex, sw = compare(ENTRIES)
    assert s_all == 0 and s_ex > 40 and n_ex < n_all and sw["flattened"] == 0    # a Config change re-flattens nothing
    n_back, _, sw2 = compare(None)
    assert n_back == n_all and sw2["flattened"] == 0


def test_review_default_ns_row():
    """pkg/webhook/policy_test.go:512-583 (TestReviewDefaultNS): Config match {excludedNamespaces: ["default"], processes: ["*"]};
    a Pod whose own metadata.namespace is "" arrives with request.namespace "default" -> the webhook allows it without a review
    (the request's namespace decides, `*` covers the webhook process)"""
    entries = [{"excludedNamespaces": ["default"], "processes": ["*"]}]
    ox = OX.Excluder(entries)
    c = make_client("hostemu")
    c.SetExcluder(entries)
    req = {"uid": "u", "kind": {"group": "", "version": "v1", "kind": "Pod"}, "userInfo": {}, "operation": "CREATE", "namespace": "default",
           "object": {"apiVersion": "v1", "kind": "Pod", "metadata": {"name": "acbd", "namespace": ""}}}
    assert ox.webhook_skips("webhook", req) is True
    assert c.IsNamespaceExcluded("webhook", D.AdmissionRequest(req)) is True
    other = dict(req, namespace="kube-public")
    assert ox.webhook_skips("webhook", other) is False and c.IsNamespaceExcluded("webhook", D.AdmissionRequest(other)) is False
