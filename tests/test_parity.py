"""Parity of the product path (AOT compiler -> flattener -> kernels -> host renderer, through the C ABI) with the
oracle, on the reference's own fixtures and on seeded synthetic workloads.  Bit-exact: violation multisets
(constraint, msg, details, enforcementAction) must be identical.

Every test runs twice: `hostemu` (CPU container; the kernels' vm_core.hpp code executed lane by lane by a test-only
library) and `gpu` (-m gpu: the HIP kernels on a real MI355X)."""
import numpy as np
import json
import os

import pytest

import reference_tables as T
from conftest import gconst, ydocs
from gatekeeper_amd import _lib as L
from gatekeeper_amd import driver as D
from gatekeeper_amd import synth
from parity_util import to_oracle_review, BACKENDS, assert_parity, key, load_both, make_client

PSP = "pkg/webhook/testdata/psp-all-violations/"


def _dir(fixtures, prefix):
    return [ydocs(fixtures, p)[0] for p in sorted(fixtures["yaml"]) if p.startswith(prefix)]


@pytest.mark.parametrize("backend", BACKENDS)
def test_psp_all_violations(backend, fixtures):
    """pkg/webhook/policy_benchmark_test.go:264-271 fixtures: 5 templates x 5 constraints x 5 pods."""
    c, oc = load_both(backend, _dir(fixtures, PSP + "psp-templates/"), _dir(fixtures, PSP + "psp-constraints/"))
    pods = _dir(fixtures, PSP + "psp-pods/")
    n = assert_parity(c, oc, [D.AugmentedUnstructured(D.Unstructured(p), None, "Original") for p in pods])
    assert n >= 5


@pytest.mark.parametrize("backend", BACKENDS)
def test_config1_demo_basic(backend, fixtures):
    d = "demo/basic/"
    c, oc = load_both(backend, ydocs(fixtures, d + "templates/k8srequiredlabels_template.yaml"),
                      ydocs(fixtures, d + "constraints/all_ns_must_have_gatekeeper.yaml"))
    objs = ydocs(fixtures, d + "bad/bad_ns.yaml") + ydocs(fixtures, d + "good/good_ns.yaml")
    got = c.ReviewBatch([D.AugmentedUnstructured(D.Unstructured(o), None, "Original") for o in objs], D.GATOR_EP)
    assert [[r.msg for r in g] for g in got] == [[T.MSG_REQUIRED_LABELS_GATEKEEPER], []]
    assert got[0][0].metadata == {"details": {"missing_labels": ["gatekeeper"]}}
    # BASELINE.json configs[0]: 1 constraint x 10 Pods (5 labelled / 5 not) -> exactly 5 violations
    cons = {"apiVersion": "constraints.gatekeeper.sh/v1beta1", "kind": "K8sRequiredLabels", "metadata": {"name": "pods-gk"},
            "spec": {"match": {"kinds": [{"apiGroups": [""], "kinds": ["Pod"]}]}, "parameters": {"labels": ["gatekeeper"]}}}
    c2, oc2 = load_both(backend, ydocs(fixtures, d + "templates/k8srequiredlabels_template.yaml"), [cons])
    pods = [{"apiVersion": "v1", "kind": "Pod", "metadata": {"name": "p%d" % i, "namespace": "default",
                                                             "labels": ({"gatekeeper": "x"} if i % 2 else {"other": "y"})}} for i in range(10)]
    rv = [D.AugmentedUnstructured(D.Unstructured(p), None, "Original") for p in pods]
    assert assert_parity(c2, oc2, rv, D.GATOR_EP) == 5


@pytest.mark.parametrize("backend", BACKENDS)
@pytest.mark.parametrize("mixed,n", [(False, 600), (True, 600)])
def test_synthetic_parity(backend, mixed, n, fixtures):
    """configs[1]/[2] at oracle-sized N: 30 PSP constraints x Pods, 50 constraints x mixed objects."""
    cons = synth.audit_constraints() if mixed else synth.psp_constraints()
    c, oc = load_both(backend, synth.psp_templates(fixtures), cons)
    nss = synth.gen_namespaces()
    objs = synth.gen_objects(n, mixed=mixed)
    rv = [D.AugmentedUnstructured(D.Unstructured(o), synth.namespace_for(o, nss), "Original") for o in objs]
    total = assert_parity(c, oc, rv)
    assert total > n // 10


@pytest.mark.parametrize("backend", BACKENDS)
def test_match_table(backend, fixtures):
    """pkg/mutation/match/match_test.go:17-684 through the device match program (deny-all template)."""
    tmpl = fixtures["go_consts"]["pkg/target/target_integration_test.go"]["testTemplate"]["docs"][0]
    for name, obj, mt, ns, source, want, want_err in T.MATCH_CASES:
        if obj is None:
            continue   # nil-object row: unreachable through HandleReview
        cons = {"apiVersion": "constraints.gatekeeper.sh/v1beta1", "kind": "DenyAll", "metadata": {"name": "c"}, "spec": {"match": mt}}
        c = make_client(backend)
        c.AddTemplate(tmpl)
        c.AddConstraint(cons, validate=False)   # match_test.go drives match.Matches directly, incl. invalid selectors
        res = c.Review(D.AugmentedUnstructured(D.Unstructured(obj), ns, source), D.AUDIT_EP)
        if want_err:
            assert len(res) == 1 and res[0].msg.startswith("unable to match constraints: "), name
        else:
            assert (len(res) == 1) is want, name
            if want:
                assert res[0].msg == "denyall constraint installed"


@pytest.mark.parametrize("backend", BACKENDS)
def test_matcher_match_table(backend, fixtures):
    """pkg/target/target_test.go:657-981 (TestMatcher_Match): review shapes, object / oldObject combinations, cached
    Namespace fallback (matcher.go:37-39) and the two error kinds (ErrMatching, ErrRequestObject -- incl. the object that
    Unstructured.UnmarshalJSON rejects, matcher.go:73-93), through the device path and through the oracle."""
    tmpl = fixtures["go_consts"]["pkg/target/target_integration_test.go"]["testTemplate"]["docs"][0]
    for name, shape, body, ns, cached, mt, want, want_err in T.MATCHER_MATCH_CASES:
        cons = {"apiVersion": "constraints.gatekeeper.sh/v1beta1", "kind": "DenyAll", "metadata": {"name": "c"}, "spec": {"match": mt}}
        c, oc = load_both(backend, [tmpl], [cons], [cached] if cached else [])
        if shape == "object":
            rv = D.AugmentedUnstructured(D.Unstructured(body), ns, "")
        else:
            rv = D.AugmentedReview(D.AdmissionRequest(dict(body)), ns, "")
        for client, review in ((c, rv), (oc, to_oracle_review(rv))):
            res = client.Review(review, D.AUDIT_EP) if client is c else client.review(review, D.AUDIT_EP, None)
            if want_err:
                assert len(res) == 1 and res[0].msg.startswith("unable to match constraints: "), (name, [r.msg for r in res])
            else:
                assert [r.msg for r in res] == (["denyall constraint installed"] if want else []), (name, [r.msg for r in res])
        assert_parity(c, oc, [rv])


def _random_match_world(seed, n_cons, n_objs):
    """Seeded random spec.match blocks (all 8 matchers, globs, selectors incl. invalid ones) and objects / namespaces."""
    rng = synth.SplitMix64(seed)
    pick = lambda xs: xs[rng.below(len(xs))]
    ns_names = ["default", "prod-1", "prod-2", "dev-1", "kube-system", "team-a", "team-b"]
    globs = ["prod-*", "*-system", "*-1", "dev-1", "team-?", "*", "*eam*", "default", "prod-2", "x*"]
    kinds = [("", "Pod"), ("", "Namespace"), ("apps", "Deployment"), ("", "Service"), ("batch", "Job")]
    label_keys, label_vals = ["env", "tier", "app", "x"], ["prod", "dev", "web", "db", ""]
    def selector():
        sel = {}
        if rng.chance(0.5):
            sel["matchLabels"] = {pick(label_keys): pick(label_vals) for _ in range(1 + rng.below(2))}
        if rng.chance(0.6):
            exprs = []
            for _ in range(1 + rng.below(2)):
                op = pick(["In", "NotIn", "Exists", "DoesNotExist", "In", "Bogus"])
                e = {"key": pick(label_keys), "operator": op}
                if op in ("In", "NotIn") or (op in ("Exists", "Bogus") and rng.chance(0.15)):
                    e["values"] = [pick(label_vals) for _ in range(rng.below(3))]   # may be empty (invalid for In/NotIn)
                exprs.append(e)
            sel["matchExpressions"] = exprs
        return sel
    cons = []
    for i in range(n_cons):
        m = {}
        if rng.chance(0.5):
            m["kinds"] = [{"apiGroups": [pick(["", "apps", "*", "batch"])] if rng.chance(0.8) else [],
                           "kinds": [pick(["Pod", "Deployment", "*", "Namespace", "Service"]) for _ in range(1 + rng.below(2))] if rng.chance(0.85) else []}
                          for _ in range(1 + rng.below(2))]
        if rng.chance(0.3):
            m["scope"] = pick(["*", "Cluster", "Namespaced", "Typo"])
        if rng.chance(0.35):
            m["namespaces"] = [pick(globs) for _ in range(1 + rng.below(2))]
        if rng.chance(0.3):
            m["excludedNamespaces"] = [pick(globs) for _ in range(1 + rng.below(2))]
        if rng.chance(0.35):
            m["labelSelector"] = selector()
        if rng.chance(0.3):
            m["namespaceSelector"] = selector()
        if rng.chance(0.25):
            m["name"] = pick(["web-*", "*-0", "db", "*e*", "*"])
        if rng.chance(0.2):
            m["source"] = pick(["All", "Original", "Generated", "Nope"])
        c = {"apiVersion": "constraints.gatekeeper.sh/v1beta1", "kind": "DenyAll", "metadata": {"name": "c%d" % i}}
        if m or rng.chance(0.5):
            c["spec"] = {"match": m}
        cons.append(c)
    nss = {n: {"apiVersion": "v1", "kind": "Namespace", "metadata": {"name": n, "labels": {pick(label_keys): pick(label_vals) for _ in range(rng.below(3))}}}
           for n in ns_names}
    reviews = []
    for j in range(n_objs):
        g, k = pick(kinds)
        meta = {"name": pick(["web-0", "web-1", "db", "cache-0", "e"])}
        if rng.chance(0.15):
            meta = {"generateName": pick(["web-", "db-", "e"])}
        if rng.chance(0.7):
            meta["labels"] = {pick(label_keys): pick(label_vals) for _ in range(rng.below(3))}
        if k == "Namespace":
            meta["name"] = pick(ns_names)
        elif rng.chance(0.8):
            meta["namespace"] = pick(ns_names)
        obj = {"apiVersion": (g + "/v1") if g else "v1", "kind": k, "metadata": meta}
        ns = nss.get(meta.get("namespace")) if rng.chance(0.75) else None      # sometimes the Namespace object is missing
        src = pick(["Original", "Generated", "", "Original"])
        reviews.append((D.AugmentedUnstructured(D.Unstructured(obj), ns, src), ns))
    return cons, reviews


@pytest.mark.parametrize("backend", BACKENDS)
def test_match_randomised(backend, fixtures):
    """Differential test of the compiled Match layer (pkg/mutation/match/match.go:32-258 incl. its error paths and the
    autoreject messages) against the oracle on seeded random match blocks x objects."""
    tmpl = fixtures["go_consts"]["pkg/target/target_integration_test.go"]["testTemplate"]["docs"][0]
    for seed in (101, 202):
        cons, reviews = _random_match_world(seed, 60, 150)
        c, oc = load_both(backend, [tmpl], cons, validate=False)
        total = assert_parity(c, oc, [r for r, _ in reviews])
        assert total > 100


@pytest.mark.parametrize("backend", BACKENDS)
def test_many_constraints_split_into_plan_groups(backend, fixtures):
    """More than 64 distinct match formulas do not fit one plan (one result bit each): the engine splits the constraint set
    into plan groups evaluated over the same table and appends their bitmap rows.  150 random match blocks + PSP
    constraints: parity per review, and the raw bitmap / counts / list / top-k API stay consistent across the groups."""
    tmpl = fixtures["go_consts"]["pkg/target/target_integration_test.go"]["testTemplate"]["docs"][0]
    cons, reviews = _random_match_world(77, 150, 130)
    c, oc = load_both(backend, [tmpl] + synth.psp_templates(fixtures), cons + synth.psp_constraints(), validate=False)
    revs = [r for r, _ in reviews]
    assert assert_parity(c, oc, revs) > 500
    table = c.driver.engine.create_table([D.to_review_in(r) for r in revs], keep_docs=False)
    ev = table.eval(want_match=True, want_list=True)
    assert ev.n_constraints == len(cons) + 30 == len(ev.constraint_ids) == len(set(int(x) for x in ev.constraint_ids))
    pop = np.array([sum(bin(int(w)).count("1") for w in row) for row in ev.viol])
    assert (pop == ev.counts).all() and ev.list_total == pop.sum() >= len(ev.list) > 0
    listed = [(int(ev.constraint_ids[a]), int(b)) for a, b in ev.list]     # (a capacity-bounded prefix per plan group)
    assert len(set(listed)) == len(listed) and set(listed) <= set(ev.pairs("viol"))
    assert {cid for cid, _ in listed} & {int(x) for x in ev.constraint_ids[64:]}, "no list entries from the later plan groups"
    assert ((ev.viol & ~ev.match) == 0).all()
    top = table.topk(5)
    for i, cid in enumerate(ev.constraint_ids):
        got, ovf = top[int(cid)]
        assert not ovf and len(got) == min(5, int(ev.counts[i])) or len(got) >= min(5, int(ev.counts[i]))
        assert all((int(ev.viol[i][r // 64]) >> (r % 64)) & 1 for r in got)
    table.free()


@pytest.mark.parametrize("backend", BACKENDS)
def test_constraint_enforcement(backend, fixtures):
    """pkg/target/target_integration_test.go:163-527: 26 scenarios x 3 review shapes, batched into one launch each."""
    tmpl = fixtures["go_consts"]["pkg/target/target_integration_test.go"]["testTemplate"]["docs"][0]
    for name, obj, ns, mt, allowed in T.ENFORCEMENT_CASES:
        cons = {"apiVersion": "constraints.gatekeeper.sh/v1beta1", "kind": "DenyAll", "metadata": {"name": "my-constraint"}}
        if mt is not None:
            cons["spec"] = {"match": mt}
        c, oc = load_both(backend, [tmpl], [cons])
        g, v = obj["apiVersion"].split("/")[0], ""
        kind = {"group": g, "version": v, "kind": obj["kind"]}
        req = {"kind": kind, "object": obj}
        req2 = {"kind": kind, "oldObject": obj}
        if ns is not None:
            req["namespace"] = req2["namespace"] = ns["metadata"]["name"]
        shapes = [D.AugmentedReview(D.AdmissionRequest(req), ns), D.AugmentedReview(D.AdmissionRequest(req2), ns),
                  D.AugmentedUnstructured(D.Unstructured(obj), ns)]
        got = c.ReviewBatch(shapes, D.AUDIT_EP)
        for g_ in got:
            assert (len(g_) == 0) is allowed, name
        assert_parity(c, oc, shapes)


@pytest.mark.parametrize("backend", BACKENDS)
def test_gator_fixtures(backend, fixtures):
    """pkg/gator/test/test_test.go:86-330 message and enforcement-point pins, via Client.Review."""
    def docs(*names):
        out = []
        for n in names:
            out.extend(gconst(fixtures, n))
        return out

    objs = docs("TemplateNeverValidate", "ConstraintNeverValidate", "Object")
    c, oc = load_both(backend, [objs[0]], [objs[1]])
    rv = [D.AugmentedUnstructured(D.Unstructured(o), None, "Original") for o in objs]
    got = c.ReviewBatch(rv, D.GATOR_EP)
    assert [r.msg for g in got for r in g] == [T.MSG_NEVER_VALIDATE] * 3
    assert_parity(c, oc, rv, D.GATOR_EP)
    # scoped enforcement actions
    objs = docs("TemplateNeverValidate", "ConstraintGatorValidate", "ConstraintAuditValidate", "Object")
    c, oc = load_both(backend, [objs[0]], [objs[1], objs[2]])
    rv = [D.AugmentedUnstructured(D.Unstructured(objs[3]), None, "Original")]
    res = c.Review(rv[0], D.GATOR_EP)
    assert [(r.enforcement_action, r.scoped_enforcement_actions) for r in res] == [("scoped", ["deny"])]
    assert_parity(c, oc, rv, D.GATOR_EP)
    assert_parity(c, oc, rv, D.AUDIT_EP)
    assert_parity(c, oc, rv, D.WEBHOOK_EP)
    # two bodies -> two messages; userInfo; restricted custom field
    for tn, cn, on in [("TemplateNeverValidateTwice", "ConstraintNeverValidateTwice", "Object"),
                       ("TemplateAlwaysValidate", "ConstraintAlwaysValidate", "Object"),
                       ("TemplateRestrictCustomField", "ConstraintRestrictCustomField", "ObjectFooTemplate")]:
        t_, k_, o_ = docs(tn, cn, on)
        c, oc = load_both(backend, [t_], [k_])
        assert_parity(c, oc, [D.AugmentedUnstructured(D.Unstructured(o_), None, "Original")], D.GATOR_EP)
    t_, k_ = docs("TemplateValidateUserInfo", "ConstraintAlwaysValidateUserInfo")
    c, oc = load_both(backend, [t_], [k_])
    for ar in ("SystemAdmissionReview", "NonSystemAdmissionReview", "AdmissionReviewWithOldObject"):
        req = gconst(fixtures, ar)[0]["request"]
        assert_parity(c, oc, [D.AugmentedReview(D.AdmissionRequest(req), None, "Original")], D.GATOR_EP)


@pytest.mark.parametrize("backend", BACKENDS)
def test_namespace_cache_and_autoreject(backend, fixtures):
    """matcher.go:37-39 nsCache fallback + the autoreject message pinned at test/gator/test/test.bats:301."""
    t_ = gconst(fixtures, "TemplateNeverValidate")[0]
    k_ = gconst(fixtures, "ConstraintNamespaceSelector")[0]
    obj = gconst(fixtures, "ObjectNamespaceScope")[0]
    for inv, want in ((["NamespaceSelected"], 1), (["NamespaceNotSelected"], 0), ([], 1)):
        c, oc = load_both(backend, [t_], [k_], [gconst(fixtures, i)[0] for i in inv])
        rv = [D.AugmentedUnstructured(D.Unstructured(obj), None, "Original")]
        res = c.Review(rv[0], D.GATOR_EP)
        assert len(res) == want
        if not inv:
            assert res[0].msg == ("unable to match constraints: error matching the requested object: object :failed to run "
                                  "Match criteria: namespace selector for namespace-scoped object but missing Namespace")
        assert_parity(c, oc, rv, D.GATOR_EP)
    # target_test.go:983-1152 (TestNamespaceCache) through Client.AddData / RemoveData: removal, non-Namespace data is
    # ignored by the cache, a Namespace that does not convert into the typed object is refused
    c, oc = load_both(backend, [t_], [k_], [gconst(fixtures, "NamespaceSelected")[0]])
    rv = [D.AugmentedUnstructured(D.Unstructured(obj), None, "Original")]
    before = [r.msg for r in c.Review(rv[0], D.GATOR_EP)]
    c.RemoveData(gconst(fixtures, "NamespaceSelected")[0])
    oc.remove_data(gconst(fixtures, "NamespaceSelected")[0])
    after = [r.msg for r in c.Review(rv[0], D.GATOR_EP)]
    assert before != after and after[0].startswith("unable to match constraints: ")
    assert_parity(c, oc, rv, D.GATOR_EP)
    with pytest.raises(D.ClientError):
        c.AddData({"apiVersion": "v1", "kind": "Namespace", "spec": 3.0})
    c.AddData({"apiVersion": "v1", "kind": "ConfigMap", "metadata": {"name": "cm", "namespace": "default"}})
    assert [r.msg for r in c.Review(rv[0], D.GATOR_EP)] == after


@pytest.mark.parametrize("backend", BACKENDS)
def test_template_families(backend, fixtures):
    """In-tree template families beyond PSP that the device plan supports."""
    d = "demo/agilebank/"
    tmpls = {p: ydocs(fixtures, p)[0] for p in fixtures["yaml"] if p.startswith(d + "templates/")}
    cons = {p: ydocs(fixtures, p)[0] for p in fixtures["yaml"] if p.startswith(d + "constraints/")}
    objs = []
    for p in sorted(fixtures["yaml"]):
        if p.startswith(d + "good_resources/") or p.startswith(d + "bad_resources/"):
            objs.extend(ydocs(fixtures, p))
    pods = synth.gen_objects(200, seed=7)
    ns = {"apiVersion": "v1", "kind": "Namespace", "metadata": {"name": "production"}}
    supported = 0
    for tp, t_ in sorted(tmpls.items()):
        kind = t_["spec"]["crd"]["spec"]["names"]["kind"]
        ks = [k for k in cons.values() if k["kind"] == kind]
        try:
            c, oc = load_both(backend, [t_], ks, [ns])
        except D.UnsupportedError:
            continue
        supported += 1
        rv = [D.AugmentedUnstructured(D.Unstructured(o), None, "Original") for o in objs + pods]
        assert_parity(c, oc, rv)
    assert supported >= 4   # requiredlabels (allowedRegex), allowedrepos, requiredprobes, containerlimits (dictionary predicates)


@pytest.mark.parametrize("backend", BACKENDS)
def test_audit_aggregation(backend, fixtures):
    """Row a11: totals + the 20 smallest violations per constraint (pkg/audit/manager.go:885-941, LimitQueue order), with
    the top-k candidates selected on the device, against the oracle's restatement fed with EVERY result.  Includes objects
    that share one (gvk, namespace, name) key so the k-th key has ties."""
    from oracle import audit as OA
    from parity_util import to_oracle_review
    c, oc = load_both(backend, synth.psp_templates(fixtures), synth.audit_constraints())
    nss = synth.gen_namespaces()
    objs = synth.gen_objects(700, seed=23, mixed=True)
    dup = json.loads(json.dumps(objs[5]))
    dup["spec"]["hostNetwork"] = True
    objs += [dup, json.loads(json.dumps(dup)), json.loads(json.dumps(objs[5]))]      # same object key, different violations
    reviews = [D.AugmentedUnstructured(D.Unstructured(o), synth.namespace_for(o, nss), "Original") for o in objs]
    for limit in (20, 3):
        got = c.AuditAggregate(reviews, limit=limit)
        lists, totals, per_action, pairs = {}, {}, {}, {}
        for o, rv in zip(objs, reviews):
            res = oc.review(to_oracle_review(rv), D.AUDIT_EP, None)
            OA.add_audit_responses(lists, totals, per_action, [(r, o) for r in res], limit=limit)
            for k in {(r.constraint.get("kind", ""), r.constraint.get("apiVersion", ""), r.constraint["metadata"]["name"]) for r in res}:
                pairs[k] = pairs.get(k, 0) + 1
        assert {k: v["total_pairs"] for k, v in got.items() if v["total_pairs"]} == pairs
        # totalViolationsPerConstraint / PerEnforcementAction count RESULTS (manager.go:902-904), not violating pairs
        assert {k: v["total"] for k, v in got.items() if v["total"]} == totals
        assert got.totals_per_action == per_action and not got.errors
        assert sum(totals.values()) > sum(pairs.values())   # the workload has multi-result pairs, so the two differ
        for k, q in lists.items():
            assert got[k]["violations"] == q.sorted(), k
            # what updateConstraintStatus writes (manager.go:980-1034): popped from the max-heap = descending order
            st = got.constraint_status(k, "2026-01-01T00:00:00Z", violations_limit=limit)
            assert st == OA.constraint_status(q, totals[k], "2026-01-01T00:00:00Z", violations_limit=limit), k
            assert [OA.sv_key(v) for v in st["violations"]] == sorted((OA.sv_key(v) for v in st["violations"]), reverse=True)
            assert all("namespace" in v or not v.get("namespace") for v in st["violations"]) and st["totalViolations"] == totals[k]
        clean = ("K8sNope", "constraints.gatekeeper.sh/v1beta1", "none")
        assert got.constraint_status(clean, "t") == {"auditTimestamp": "t", "totalViolations": 0}     # status.violations removed
        assert sum(len(v["violations"]) for v in got.values()) == sum(len(q.items) for q in lists.values()) > 0


def _mutate(rng, v):
    """random structural mutation of a JSON value: dropped / added members, duplicated elements, subtrees replaced by
    scalars of the wrong type, numeric edge cases, long and non-ASCII strings"""
    pick = lambda xs: xs[rng.below(len(xs))]
    scalars = [None, True, False, 0, 1, -1, 80, 65535, 65536, 9000.5, 1e3, 2 ** 53, 2 ** 63, -2 ** 63, "", "x", "true", "privileged", "/foo/bar",
               "hostPath", "a-very-long-string-value-exceeding-twelve-bytes", "\u00fcn\u00efc\u00f6d\u00e9", "0", "80", [], {}, [1, "a"], {"a": 1}]
    if rng.chance(0.08):
        return pick(scalars)
    if isinstance(v, dict):
        out = {k: _mutate(rng, x) for k, x in v.items() if not rng.chance(0.04)}
        if rng.chance(0.05):
            out[pick(["extra", "name", "hostPath", "privileged", "readOnly", "x-y"])] = pick(scalars)
        return out
    if isinstance(v, list):
        out = [_mutate(rng, x) for x in v if not rng.chance(0.05)]
        if out and rng.chance(0.1):
            out.append(json.loads(json.dumps(out[0])))
        if rng.chance(0.03):
            out.append(pick(scalars))
        return out
    return v


@pytest.mark.parametrize("backend", BACKENDS)
def test_structural_fuzz(backend, fixtures):
    """Seeded structural fuzzing of the reviewed objects (wrong types, missing members, arrays where objects are expected
    -- e.g. key iteration over an array yields numeric keys) through flattener, compiled predicates and renderer, against
    the oracle; object and AdmissionRequest (CREATE / UPDATE / DELETE with oldObject) shapes."""
    nss = synth.gen_namespaces()
    for seed, caps in ((1, None), (22, (2, 3, 2))):   # tiny element capacities: most reviews take the large-variant kernel
        c, oc = load_both(backend, synth.psp_templates(fixtures), synth.audit_constraints(), **({"elem_cap": caps} if caps else {}))
        rng = synth.SplitMix64(seed)
        revs = []
        for o in synth.gen_objects(220, seed=seed, mixed=True):
            m = _mutate(rng, _mutate(rng, o))
            if not isinstance(m, dict):
                m = {"apiVersion": "v1", "kind": "Pod", "metadata": {"name": "m"}}
            if not (isinstance(m.get("kind"), str) and m["kind"]):
                m["kind"] = "Pod"      # objects without a kind are the known gap pinned by test_matcher_match_table
            md = m.get("metadata")
            ns = synth.namespace_for(m, nss) if isinstance(md, dict) and isinstance(md.get("namespace"), str) else None
            shape = rng.below(4)
            if shape == 0:
                revs.append(D.AugmentedUnstructured(D.Unstructured(m), ns, "Original"))
                continue
            api = m.get("apiVersion") if isinstance(m.get("apiVersion"), str) else "v1"
            g, _, ver = api.rpartition("/")
            req = {"kind": {"group": g, "version": ver, "kind": m.get("kind") if isinstance(m.get("kind"), str) else ""},
                   "operation": ["CREATE", "UPDATE", "DELETE"][shape - 1]}
            if isinstance(md, dict) and isinstance(md.get("namespace"), str):
                req["namespace"] = md["namespace"]
            if shape == 1:
                req["object"] = m
            elif shape == 2:
                old = _mutate(rng, m)
                if not (isinstance(old, dict) and isinstance(old.get("kind"), str) and old["kind"]):
                    old = m
                req["object"], req["oldObject"] = m, old
            else:
                req["oldObject"] = m
            revs.append(D.AugmentedReview(D.AdmissionRequest(req), ns, "Original"))
        refused = []
        assert assert_parity(c, oc, revs, refused=refused) > 50
        assert len(refused) < len(revs) // 10     # (objects mutated into the place of an iterated array: refused, fail closed)


REGEX_TEMPLATE = {
    "apiVersion": "templates.gatekeeper.sh/v1", "kind": "ConstraintTemplate", "metadata": {"name": "k8slabelregex"},
    "spec": {"crd": {"spec": {"names": {"kind": "K8sLabelRegex"}}},
             "targets": [{"target": "admission.k8s.gatekeeper.sh", "rego": """
package k8slabelregex
violation[{"msg": msg}] {
  value := input.review.object.metadata.labels[key]
  expected := input.parameters.labels[_]
  expected.key == key
  expected.allowedRegex != ""
  not re_match(expected.allowedRegex, value)
  msg := sprintf("Label <%v: %v> does not satisfy allowed regex: %v", [key, value, expected.allowedRegex])
}
violation[{"msg": msg}] {
  re_match(input.parameters.nameRegex, input.review.object.metadata.name)
  msg := sprintf("name %v matches %v", [input.review.object.metadata.name, input.parameters.nameRegex])
}
"""}]}}

REGEXES = ["^[a-zA-Z]+.agilebank.demo$", "^team-[0-9]+$", "prod|dev", "^(a|b)*c$", "[[:digit:]]{2,3}", "^$", "a.c", "^x?y+z*$",
           "\\d+\\.\\d+", "(foo|bar)baz$", "^[^0-9]*$", ".", "^web-[a-f0-9]{4,}-(blue|green)$", "(", "kube-system-extended-name-[0-9]+"]
REGEX_VALUES = ["", "a", "abc", "axc", "team-7", "team-", "xteam-12x", "prod", "devops", "aabbc", "abab", "42", "1234", "1.5", "v1.25.3",
                "foobaz", "barbazz", "xyz", "yyzz", "y", "user.agilebank.demo", "user1.agilebank.demo", "web-0a1f-blue", "web-0a1-green",
                "web-deadbeef-green-canary", "kube-system-extended-name-0042", "a\nc", "long-label-value-that-goes-well-beyond-twelve-bytes-77"]


@pytest.mark.parametrize("backend", BACKENDS)
def test_regex_predicates(backend):
    """re_match(constant pattern, review string) compiles to a DFA predicate: differential test against the oracle over
    anchors, classes, alternation, repetition, an invalid pattern (undefined), inline and heap strings."""
    cons = [{"apiVersion": "constraints.gatekeeper.sh/v1beta1", "kind": "K8sLabelRegex", "metadata": {"name": "re-%d" % i},
             "spec": {"parameters": {"labels": [{"key": "owner", "allowedRegex": rx}, {"key": "tier", "allowedRegex": REGEXES[(i + 3) % len(REGEXES)]}],
                                     "nameRegex": REGEXES[(i + 7) % len(REGEXES)]}}}
            for i, rx in enumerate(REGEXES)]
    c, oc = load_both(backend, [REGEX_TEMPLATE], cons)
    objs = [{"apiVersion": "v1", "kind": "Pod", "metadata": {"name": REGEX_VALUES[(i * 5 + 2) % len(REGEX_VALUES)] or "x", "namespace": "default",
                                                           "labels": {"owner": v, "tier": REGEX_VALUES[(i * 3 + 1) % len(REGEX_VALUES)]}}}
            for i, v in enumerate(REGEX_VALUES)]
    objs.append({"apiVersion": "v1", "kind": "Pod", "metadata": {"name": "nolabels", "namespace": "default"}})
    objs.append({"apiVersion": "v1", "kind": "Pod", "metadata": {"name": "numlabel", "namespace": "default", "labels": {"owner": 5}}})
    assert_parity(c, oc, [D.AugmentedUnstructured(D.Unstructured(o), None, "Original") for o in objs])


GO_REGEXES = ["(?i)^PROD", "\\bprod\\b", "^.{3}$", "^[^a]+$", "\\x41\\x{e9}", "\\Qa.b\\E+", "(?s)a.c", "a.c", "^\\s+$", "\\S\\s\\S", "(?i)k+", "(?i)[j-l]s$",
              "a**", "x{2,1}", "\\8", "(?P<n>ab)+", "(?i:A)b", "a(?i)b|c", "(?m)^c$", "^c$", "\\Bb", "\\W", "[[:upper:]][[:^digit:]]", "\\pL+", "[\u00e9x]y",
              "(?i)\u00e9", "[^\\x00-\\x7f]", "\\D\\d", "a{,2}", "\\z|^q", "\\101", "(?U)a+?b"]
GO_VALUES = ["prod", "PROD-1", "bprodb", "a prod b", "h\u00e9\u00e9", "abc", "A\u00e9", "a.b.b", "a.bb", "a\nc", "axc", " \t\n", "\x0b", "x y", "KK", "k\u212a",
             "LS", "l\u017f", "ab", "abab", "Ab", "aB", "C", "a\nc\nd", "c", "ab b", "b", "-", "Q7", "QZ", "\u00e9y", "\u00c9", "\u00e9", "5", "a5", "a{,2}", "q", "A",
             "aab", "\u65e5\u672c", "x\U0001f600y"]


@pytest.mark.parametrize("backend", BACKENDS)
def test_regex_go_syntax(backend):
    """Go regexp/syntax constructs beyond the reference's fixtures: inline flags, word boundaries, hex / octal /
    \\Q..\\E escapes, rune-wise '.' and negated classes on non-ASCII text, case folding onto U+212A / U+017F, patterns
    regexp.Compile rejects (re_match undefined), and valid patterns outside the engine (\\pL: GK_ERR_UNSUPPORTED at
    AddConstraint, never a silent mis-evaluation).  Oracle: the restated Go->Python translation (oracle/rego_builtins.py)."""
    from oracle import rego_builtins as OB
    usable, rejected = [], []
    for rx in GO_REGEXES:
        try:
            OB.go_regex(rx)
        except OB.OracleRegexUnsupported:
            oracle_ok = False
        except OB.BuiltinError:
            oracle_ok = True    # invalid in Go: both sides must treat re_match as undefined
        else:
            oracle_ok = True
        probe = make_client(backend)
        probe.AddTemplate(REGEX_TEMPLATE)
        try:
            probe.AddConstraint({"apiVersion": "constraints.gatekeeper.sh/v1beta1", "kind": "K8sLabelRegex", "metadata": {"name": "t"},
                                 "spec": {"parameters": {"labels": [{"key": "owner", "allowedRegex": rx}], "nameRegex": rx}}})
        except D.UnsupportedError:
            rejected.append(rx)
            continue
        if oracle_ok:
            usable.append(rx)
    assert "\\pL+" in rejected and "(?i)\u00e9" in rejected          # valid Go, outside the engine: reported, not approximated
    assert len(usable) >= len(GO_REGEXES) - 6, rejected
    cons = [{"apiVersion": "constraints.gatekeeper.sh/v1beta1", "kind": "K8sLabelRegex", "metadata": {"name": "go-%d" % i},
             "spec": {"parameters": {"labels": [{"key": "owner", "allowedRegex": rx}], "nameRegex": usable[(i + 5) % len(usable)]}}}
            for i, rx in enumerate(usable)]
    c, oc = load_both(backend, [REGEX_TEMPLATE], cons)
    objs = [{"apiVersion": "v1", "kind": "Pod", "metadata": {"name": GO_VALUES[(i * 7 + 3) % len(GO_VALUES)], "namespace": "default", "labels": {"owner": v}}}
            for i, v in enumerate(GO_VALUES)]
    assert assert_parity(c, oc, [D.AugmentedUnstructured(D.Unstructured(o), None, "Original") for o in objs]) > len(GO_VALUES)


def _random_regex(rng, depth=0):
    pick = lambda xs: xs[rng.below(len(xs))]
    atoms = ["a", "b", "c", "0", "1", "-", ".", "[ab]", "[^a]", "[0-9]", "\\d", "\\w", "[a-c0-1]", "x", "\\b", "\\s", "\u00e9", "(?i)", "k", "\\W", "\\x41",
             "(?i:b)", "(?s).", "[^\\d-]"]

    def piece():
        a = "(" + _random_regex(rng, depth + 1) + ")" if depth < 2 and rng.chance(0.25) else pick(atoms)
        return a + pick(["", "", "", "*", "+", "?", "{2}", "{1,2}", "{0,1}", "*?"])

    def seq():
        return "".join(piece() for _ in range(1 + rng.below(4)))
    r = seq()
    while rng.chance(0.25):
        r += "|" + seq()
    if depth == 0:
        r = ("^" if rng.chance(0.4) else "") + r + ("$" if rng.chance(0.4) else "")
    return r


@pytest.mark.parametrize("backend", BACKENDS)
def test_regex_randomised(backend):
    """Seeded random patterns (groups, alternation, classes, bounded and lazy repetition, anchors) x random strings: the
    DFA predicate on the device path against the oracle."""
    rng = synth.SplitMix64(5)
    pats = []
    while len(pats) < 30:
        rx = _random_regex(rng)
        probe = make_client(backend)
        probe.AddTemplate(REGEX_TEMPLATE)
        try:
            probe.AddConstraint({"apiVersion": "constraints.gatekeeper.sh/v1beta1", "kind": "K8sLabelRegex", "metadata": {"name": "t"},
                                 "spec": {"parameters": {"labels": [{"key": "owner", "allowedRegex": rx}], "nameRegex": rx}}})
            pats.append(rx)
        except D.UnsupportedError:
            pass   # automaton above the device limit of 255 states: reported, not approximated
    cons = [{"apiVersion": "constraints.gatekeeper.sh/v1beta1", "kind": "K8sLabelRegex", "metadata": {"name": "re-%d" % i},
             "spec": {"parameters": {"labels": [{"key": "owner", "allowedRegex": rx}], "nameRegex": pats[(i + 7) % len(pats)]}}}
            for i, rx in enumerate(pats)]
    c, oc = load_both(backend, [REGEX_TEMPLATE], cons)
    al = "abc01-x.AK \n\u00e9\u212a"
    rs = lambda: "".join(al[rng.below(len(al))] for _ in range(rng.below(16)))
    objs = [{"apiVersion": "v1", "kind": "Pod", "metadata": {"name": rs() or "n", "namespace": "default", "labels": {"owner": rs()}}} for _ in range(100)]
    assert assert_parity(c, oc, [D.AugmentedUnstructured(D.Unstructured(o), None, "Original") for o in objs]) > 100


@pytest.mark.parametrize("backend", BACKENDS)
def test_gator_verify_suite_template(backend, fixtures):
    """test/gator/verify/{template,constraint*,allow_foo,deny_foo}.yaml (the reference's own `gator verify` suite): the
    template reads its input through object.get(input, "parameters", {}) / object.get(input.review.object, "foo", "")."""
    d = "test/gator/verify/"
    tmpl = ydocs(fixtures, d + "template.yaml")[0]
    objs = [ydocs(fixtures, d + "allow_foo.yaml")[0], ydocs(fixtures, d + "deny_foo.yaml")[0],
            {"apiVersion": "v1", "kind": "Object", "metadata": {"name": "nofoo"}}, {"apiVersion": "v1", "kind": "Object", "metadata": {"name": "n"}, "foo": 5}]
    for cfile, ep in (("constraint.yaml", D.GATOR_EP), ("constraint_with_scopedEA.yaml", D.GATOR_EP),
                      ("constraint_with_scopedEA_without_gator_ep.yaml", D.GATOR_EP), ("constraint.yaml", D.AUDIT_EP)):
        c, oc = load_both(backend, [tmpl], [ydocs(fixtures, d + cfile)[0]])
        rv = [D.AugmentedUnstructured(D.Unstructured(o), None, "Original") for o in objs]
        got = c.ReviewBatch(rv, ep)
        if cfile == "constraint.yaml":
            assert len(got[0]) == 0 and len(got[1]) == 1 and "but want" in got[1][0].msg
        assert_parity(c, oc, rv, ep)


@pytest.mark.parametrize("backend", BACKENDS)
def test_unsupported_is_an_error_not_a_fallback(backend, fixtures):
    # valid Rego that no plan expresses (sorting review data in the rule body itself): refused, never approximated
    t_ = {"apiVersion": "templates.gatekeeper.sh/v1", "kind": "ConstraintTemplate", "metadata": {"name": "k8ssorted"},
          "spec": {"crd": {"spec": {"names": {"kind": "K8sSorted"}}}, "targets": [{"target": "admission.k8s.gatekeeper.sh", "rego":
              'package k\nviolation[{"msg": "m"}] { sort(input.review.object.spec.names)[0] == input.parameters.first }\n'}]}}
    k_ = {"apiVersion": "constraints.gatekeeper.sh/v1beta1", "kind": "K8sSorted", "metadata": {"name": "x"}, "spec": {"parameters": {"first": "a"}}}
    c = make_client(backend)
    c.AddTemplate(t_)
    with pytest.raises(D.UnsupportedError):
        c.AddConstraint(k_)
    with pytest.raises(D.ClientError):
        make_client(backend).AddTemplate(gconst(fixtures, "TemplateCompileError")[0])


@pytest.mark.parametrize("backend", BACKENDS)
def test_edge_cases(backend, fixtures):
    """empty and ragged batches, DELETE handling, element-capacity overflow (large-variant kernel), engine limits."""
    c, oc = load_both(backend, synth.psp_templates(fixtures), synth.psp_constraints(), elem_cap=(4, 4, 4))
    assert c.ReviewBatch([]) == []
    # ragged: 1, 63, 64, 65 reviews
    objs = synth.gen_objects(65, seed=11)
    for n in (1, 63, 64, 65):
        assert_parity(c, oc, [D.AugmentedUnstructured(D.Unstructured(o), None, "Original") for o in objs[:n]])
    # overflow: a pod with 40 privileged containers / 40 hostPath volumes exceeds the LDS capacity of 4
    big = {"apiVersion": "v1", "kind": "Pod", "metadata": {"name": "big", "namespace": "prod-01"}, "spec": {
        "containers": [{"name": "c%d" % i, "image": "x", "securityContext": {"privileged": i == 37},
                        "volumeMounts": [{"name": "v%d" % i, "mountPath": "/m", "readOnly": i % 2 == 0}]} for i in range(40)],
        "volumes": [{"name": "v%d" % i, "hostPath": {"path": "/foo/x%d" % i}} for i in range(40)]}}
    def alone(o):   # does this object fit the LDS capacity of 4 by itself?
        t_ = c.driver.engine.create_table([D.to_review_in(D.AugmentedUnstructured(D.Unstructured(o), None, "Original"))])
        n_ = t_.eval().n_overflow
        t_.free()
        return n_ == 0
    small = [o for o in objs if alone(o)][:2]
    assert len(small) == 2 and not alone(big)
    rv = [D.AugmentedUnstructured(D.Unstructured(o), None, "Original") for o in [small[0], big, small[1]]]
    table = c.driver.engine.create_table([D.to_review_in(r) for r in rv])
    ev = table.eval()
    assert ev.n_overflow == 1
    table.free()
    # a resident table gets a plan variant sized for its largest arrays: the same review stays on the LDS kernel
    table = c.driver.engine.create_table([D.to_review_in(r) for r in rv], resident=True)
    ev2 = table.eval()
    # (best effort: when the variant's accumulators would not fit in LDS the default plan serves the table)
    assert ev2.n_overflow in (0, 1) and (ev2.viol == ev.viol).all() and (ev2.err == ev.err).all() and (ev2.counts == ev.counts).all()
    table.free()
    assert_parity(c, oc, rv)
    # a whole tile of very wide pods: more 64-row chunks than a wave queues in LDS -> every review of the tile takes the
    # big path (on the GPU backends; the CPU emulation has no chunk lists), results unchanged
    wide = [{"apiVersion": "v1", "kind": "Pod", "metadata": {"name": "wide-%d" % j, "namespace": "prod-01"}, "spec": {
        "containers": [{"name": "c%d" % i, "image": "x", "securityContext": {"privileged": (i + j) % 29 == 0},
                        "ports": [{"containerPort": 80, "hostPort": 8000 + i}]} for i in range(90)]}} for j in range(64)]
    wrv = [D.AugmentedUnstructured(D.Unstructured(o), None, "Original") for o in wide + objs[:3]]
    table = c.driver.engine.create_table([D.to_review_in(r) for r in wrv])
    ev = table.eval()
    assert ev.n_overflow >= 64 and int(ev.too_big.sum()) == 0
    table.free()
    assert_parity(c, oc, wrv)
    # DELETE: object := oldObject (target.go:269-287); missing oldObject is a review error
    req = {"kind": {"group": "", "version": "v1", "kind": "Pod"}, "operation": "DELETE", "oldObject": objs[2]}
    assert_parity(c, oc, [D.AugmentedReview(D.AdmissionRequest(req), None, "Original")])
    with pytest.raises(D.ClientError):
        c.Review(D.AugmentedReview(D.AdmissionRequest({"operation": "DELETE", "object": objs[2]}), None, "Original"))
    # beyond the DEVICE's limits (>255 elements of one array): never guessed by a kernel -- since round 5 answered by the engine's own
    # exact host evaluator (match layer from the stripped review on the device, violation set by the evaluator that renders the
    # messages), reported in host_evaluated instead of too_big
    huge = {"apiVersion": "v1", "kind": "Pod", "metadata": {"name": "huge"}, "spec": {
        "containers": [{"name": "c%d" % i, "image": "x"} for i in range(300)]}}
    table = c.driver.engine.create_table([D.to_review_in(D.AugmentedUnstructured(D.Unstructured(huge), None, "Original"))])
    ev = table.eval()
    assert int(ev.too_big[0]) == 0 and ev.host_evaluated == [0]
    table.free()
    assert_parity(c, oc, [D.AugmentedUnstructured(D.Unstructured(huge), None, "Original")])
    # ... and a padded array does not evade the constraints: the privileged container among 300 is found by every caller
    huge_priv = json.loads(json.dumps(huge))
    huge_priv["spec"]["containers"][7]["securityContext"] = {"privileged": True}
    hr = D.AugmentedUnstructured(D.Unstructured(huge_priv), None, "Original")
    want = sorted(key(r) for r in oc.review(to_oracle_review(hr), D.AUDIT_EP, None))
    assert want and sorted(key(r) for r in c.Review(hr)) == want
    assert sorted((r.constraint["metadata"]["name"], r.msg) for r in c.driver.QueryMatching(D.TARGET_NAME, list(c.constraints.values()), hr).results) == \
        sorted((r.constraint["metadata"]["name"], r.msg) for r in oc.review(to_oracle_review(hr), D.AUDIT_EP, None))
    batch = c.ReviewBatch([rv[0], hr, rv[1]])
    assert sorted(key(r) for r in batch[1]) == want
    assert sorted(key(r) for r in batch[0]) == sorted(key(r) for r in oc.review(to_oracle_review(rv[0]), D.AUDIT_EP, None))
    assert sorted(key(r) for r in batch[2]) == sorted(key(r) for r in oc.review(to_oracle_review(rv[1]), D.AUDIT_EP, None))
    rep = c.AuditAggregate([rv[0], hr, rv[1]])
    assert not rep.errors
    assert_parity(c, oc, [rv[0], hr, rv[1]])
    # an array of > 255 elements that NO element predicate iterates does not put the review beyond the limits:
    # one privileged container + 300 finalizers evaluates exactly (the oracle finds the same violations)
    padded = {"apiVersion": "v1", "kind": "Pod", "metadata": {"name": "padded", "namespace": "prod-01", "finalizers": ["f%d" % i for i in range(300)]},
              "spec": {"containers": [{"name": "c0", "image": "x", "securityContext": {"privileged": True}}]}}
    assert assert_parity(c, oc, [D.AugmentedUnstructured(D.Unstructured(padded), None, "Original")]) > 0


@pytest.mark.parametrize("backend", BACKENDS)
def test_bitmap_list_counts_consistency(backend, fixtures):
    """Size-independent invariants of one launch: counts == popcount(bitmap rows), list == bitmap, viol subset of match."""
    c, _ = load_both(backend, synth.psp_templates(fixtures), synth.audit_constraints())
    nss = synth.gen_namespaces()
    objs = synth.gen_objects(1000, seed=5, mixed=True)
    rins = [D.to_review_in(D.AugmentedUnstructured(D.Unstructured(o), synth.namespace_for(o, nss), "Original")) for o in objs]
    table = c.driver.engine.create_table(rins, keep_docs=False)
    ev = table.eval(want_match=True, want_list=True)
    pop = np.array([sum(bin(int(w)).count("1") for w in row) for row in ev.viol])
    assert (pop == ev.counts).all()
    assert ev.list_total == pop.sum() == len(ev.list)
    rows = {int(cid): i for i, cid in enumerate(ev.constraint_ids)}
    assert sorted((int(ev.constraint_ids[a]), int(b)) for a, b in ev.list) == ev.pairs("viol")
    assert ((ev.viol & ~ev.match) == 0).all() and ((ev.err & ev.match) == 0).all()
    # shard invariance: evaluating two halves separately gives the same bits (audit shards objects across GPUs)
    half = len(rins) // 2 // 64 * 64
    t1, t2 = c.driver.engine.create_table(rins[:half], keep_docs=False), c.driver.engine.create_table(rins[half:], keep_docs=False)
    e1, e2 = t1.eval(), t2.eval()
    assert (np.concatenate([e1.viol, e2.viol], axis=1) == ev.viol).all()
    assert (e1.counts + e2.counts == ev.counts).all()
    # host-side build: one thread == several threads (tile ranges flattened in parallel, parts appended in order)
    os.environ["GK_HOST_THREADS"] = "1"
    ts = c.driver.engine.create_table(rins, keep_docs=False)
    os.environ["GK_HOST_THREADS"] = "3"
    tp = c.driver.engine.create_table(rins, keep_docs=False)
    del os.environ["GK_HOST_THREADS"]
    es, ep = ts.eval(want_match=True), tp.eval(want_match=True)
    for x in (es, ep):
        assert (x.viol == ev.viol).all() and (x.err == ev.err).all() and (x.match == ev.match).all() and x.n_rows == ev.n_rows
    ts.free(); tp.free()
    # resident (table-specialised plan variant, smaller LDS footprint) == default plan
    tr = c.driver.engine.create_table(rins, keep_docs=False, resident=True)
    er = tr.eval(want_match=True, want_list=True)
    assert (er.viol == ev.viol).all() and (er.err == ev.err).all() and (er.match == ev.match).all() and (er.counts == ev.counts).all()
    assert sorted(map(tuple, er.list.tolist())) == sorted(map(tuple, ev.list.tolist()))
    assert er.lds_bytes <= ev.lds_bytes
    for t in (table, t1, t2, tr):
        t.free()
    assert rows


@pytest.mark.parametrize("backend", BACKENDS)
@pytest.mark.parametrize("rpt,rpp", [(128, None), (256, None), (512, None), (512, 128), (256, 64)])
def test_row_group_geometries(backend, rpt, rpp, fixtures):
    """A table's row-group size (64 / 256 / 512 reviews, fixed when it is flattened) selects the dominant kernel's geometry
    (256 / 512 / 1024 threads per group; kernel_body.inc), and a group whose accumulators do not fit in LDS is processed in
    several passes (forced here with GK_FORCE_RPP): every geometry gives the bits of the 64-review layout, which
    test_synthetic_parity pins against the oracle -- incl. a ragged last group and reviews on the large-capacity variant."""
    c, oc = load_both(backend, synth.psp_templates(fixtures), synth.audit_constraints())
    nss = synth.gen_namespaces()
    objs = synth.gen_objects(1100, seed=9, mixed=True)
    objs[700]["spec"] = {"containers": [{"name": "c%d" % i, "image": "x", "securityContext": {"privileged": i % 7 == 0},
                                         "ports": [{"containerPort": 80, "hostPort": 8000 + i}]} for i in range(30)]}   # overflows the default capacities
    rins = [D.to_review_in(D.AugmentedUnstructured(D.Unstructured(o), synth.namespace_for(o, nss), "Original")) for o in objs]
    base = c.driver.engine.create_table(rins, keep_docs=False)
    ev0 = base.eval(want_match=True, want_list=True)
    os.environ["GK_RPT"] = str(rpt)
    if rpp:
        os.environ["GK_FORCE_RPP"] = str(rpp)
    try:
        for resident in (False, True):
            t = c.driver.engine.create_table(rins, keep_docs=False, resident=resident)
            ev = t.eval(want_match=True, want_list=True)
            assert (ev.viol == ev0.viol).all() and (ev.err == ev0.err).all() and (ev.match == ev0.match).all()
            assert (ev.counts == ev0.counts).all() and (ev.too_big == ev0.too_big).all()
            assert sorted(map(tuple, ev.list.tolist())) == sorted(map(tuple, ev0.list.tolist()))
            assert (ev.n_overflow <= ev0.n_overflow) if resident else (ev.n_overflow == ev0.n_overflow >= 1)
            t.free()
    finally:
        os.environ.pop("GK_RPT", None)
        os.environ.pop("GK_FORCE_RPP", None)
    base.free()
    # and the oracle on a sample through the large geometry
    os.environ["GK_RPT"] = str(rpt)
    try:
        assert_parity(c, oc, [D.AugmentedUnstructured(D.Unstructured(o), synth.namespace_for(o, nss), "Original") for o in objs[650:760]])
    finally:
        os.environ.pop("GK_RPT", None)


@pytest.mark.parametrize("backend", BACKENDS)
def test_container_limits_dictionary_predicates(backend, fixtures):
    """K8sContainerLimits (demo/agilebank, test/bats, test/gator/policy copies): the quantity parsing -- replace / substring /
    to_number / re_match / arithmetic over a dozen function bodies -- is a pure function of ONE leaf (the limit string), so it
    runs as a DICTIONARY predicate: the flattener evaluates it per distinct value and ships a bit, the device tests the bit
    (csrc/dexpr.hpp).  Parity with the oracle incl. unparseable, numeric, empty, missing and container-typed limits."""
    paths = ["demo/agilebank/templates/k8scontainterlimits_template.yaml", "test/bats/tests/templates/k8scontainterlimits_template.yaml",
             "test/gator/policy/testdata/templates/containerlimits/template.yaml"]
    quantities = [("100m", "1Gi"), ("2", "2Gi"), (1, 1024), ("abc", "1Zi"), ("", ""), ("0.5", "500M"), ("300m", "1G"), ("200m", "1073741824"), ("1e3", "1Ei"),
                  (None, None), ("250m", "128Mi"), ("m", "Gi"), (0.1, 1.5), ("201m", "1025Mi"), ("0200m", "1024Mi"), ([1, 2], {"a": 1}), ("1", "1Ti"), ("199m", "999M"),
                  (True, False), ("2m", "1000000m"), ("\u00b5", "1Ki\u00e9")]
    weird = [{"apiVersion": "v1", "kind": "Pod", "metadata": {"name": "w%d" % i, "namespace": "default"},
              "spec": {"containers": [{"name": "c", "resources": {"limits": {"cpu": cpu, "memory": mem}}}, {"name": "nolimits"}, {"name": "half", "resources": {"limits": {"cpu": cpu}}}],
                       "initContainers": [{"name": "i", "resources": {"limits": {"memory": mem, "cpu": "1"}}}]}} for i, (cpu, mem) in enumerate(quantities)]
    pods = synth.gen_objects(300, seed=7)
    seen = 0
    for p in paths:
        if p not in fixtures["yaml"]:
            continue
        t_ = json.loads(json.dumps(ydocs(fixtures, p)[0]))
        kind = t_["spec"]["crd"]["spec"]["names"]["kind"]
        t_["metadata"]["name"] = kind.lower()   # (the gator/policy test copy deliberately carries a non-matching name)
        cons = [{"apiVersion": "constraints.gatekeeper.sh/v1beta1", "kind": kind, "metadata": {"name": "lim-%d" % i},
                 "spec": {"match": {"kinds": [{"apiGroups": [""], "kinds": ["Pod"]}]}, "parameters": prm}}
                for i, prm in enumerate([{"cpu": "200m", "memory": "1Gi"}, {"cpu": "1", "memory": "500M"}, {"cpu": "bogus", "memory": "1Xi"}])]
        c, oc = load_both(backend, [t_], cons)
        assert assert_parity(c, oc, [D.AugmentedUnstructured(D.Unstructured(o), None, "Original") for o in weird + pods]) > 100
        seen += 1
    assert seen >= 1


@pytest.mark.parametrize("backend", BACKENDS)
def test_policy_corpus_200_templates(backend, fixtures):
    """BASELINE.json configs[4] at oracle-sized N: 200 ConstraintTemplates (the in-tree families: required labels with
    allowedRegex, allowed repos, banned image tags, container limits, required probes, 5 x PSP -- every copy its own kind)
    + 200 constraints with regex allow-lists and `namespaces: ["prod-*", "*-system"]` globs, over the mixed object stream."""
    templates, cons = synth.corpus()
    assert len(templates) == 200 and len({t["spec"]["crd"]["spec"]["names"]["kind"] for t in templates}) == 200
    if backend == "hostemu-gen":   # (the second CPU backend -- the generated plan source through g++ -- takes every other copy: the CPU suite's budget;
        templates, cons = templates[::2], cons[::2]   #  hostemu and both device backends load all 200)
    c, oc = load_both(backend, templates, cons)
    nss = synth.gen_namespaces()
    objs = synth.gen_objects(64, seed=11, mixed=True)     # (the oracle's tree-walker pays ~1 ms per pair: 12 800 of them here; tests/test_cpu_ref.py
    rv = [D.AugmentedUnstructured(D.Unstructured(o), synth.namespace_for(o, nss), "Original") for o in objs]   #  takes the corpus to thousands of objects against the compiled loop)
    assert assert_parity(c, oc, rv) > (200 if backend == "hostemu-gen" else 450)


@pytest.mark.parametrize("backend", BACKENDS)
def test_member_names_and_objects_where_arrays_are_iterated(backend, fixtures):
    """(1) A member literally named like an internal marker ("\\x01[]", "[]", "$d", "$m") is an ordinary key: it aliases
    neither the array-element step of the key-path dictionary nor the synthetic subtrees.  (2) `containers[_]` over an
    OBJECT walks its values in Rego; the device plan iterates array elements only, so such a review is never answered by a
    kernel -- the engine's host evaluator answers it (round 5; until then: refused, LimitError), with what Rego says."""
    tmpl = next(t for t in synth.psp_templates(fixtures) if t["spec"]["crd"]["spec"]["names"]["kind"] == "K8sPSPPrivilegedContainer")
    cons = [{"apiVersion": "constraints.gatekeeper.sh/v1beta1", "kind": "K8sPSPPrivilegedContainer", "metadata": {"name": "p"},
             "spec": {"match": {"kinds": [{"apiGroups": [""], "kinds": ["Pod"]}]}}}]
    c, oc = load_both(backend, [tmpl], cons)
    priv = {"name": "c", "image": "i", "securityContext": {"privileged": True}}
    arrays, objects = [], []
    for i, weird in enumerate(["\x01[]", "[]", "$d", "$m", "$ns", "plain"]):
        objects.append({"apiVersion": "v1", "kind": "Pod", "metadata": {"name": "a%d" % i, "namespace": "d", "labels": {weird: "x"}},
                        "spec": {"containers": {weird: priv}, weird: [priv]}})               # containers is an OBJECT with that member
        arrays.append({"apiVersion": "v1", "kind": "Pod", "metadata": {"name": "b%d" % i, "namespace": "d", "labels": {weird: "x"}},
                       "spec": {"containers": [priv, {"name": "ok", "image": "i", weird: {"securityContext": {"privileged": True}}}], weird: {"containers": [priv]}}})
        arrays.append({"apiVersion": "v1", "kind": "Pod", "metadata": {"name": "c%d" % i, "namespace": "d"},
                       "spec": {"containers": [{"name": "ok", "image": "i", weird: [priv]}], "initContainers": {}}})   # an EMPTY object iterates nothing
    wrap = lambda p: D.AugmentedUnstructured(D.Unstructured(p), None, "Original")   # noqa: E731
    assert assert_parity(c, oc, [wrap(p) for p in arrays]) == 6
    refused = []
    assert assert_parity(c, oc, [wrap(p) for p in objects + arrays[:2]], refused=refused) == len(objects) + 1      # what Rego says: a violation each (and one of the two array-shaped pods)
    assert refused == []
    rins = [D.to_review_in(wrap(p)) for p in objects + arrays[:2]]
    table = c.driver.engine.create_table(rins, keep_docs=False)
    ev = table.eval()
    assert ev.host_evaluated == list(range(len(objects))) and not ev.too_big_reviews()
    table.free()


OPERATION_REGO = '''package k
violation[{"msg": msg}] {
  op := object.get(input.review, "operation", "<none>")
  has_obj := object.get(input.review, "object", null) != null
  has_old := object.get(input.review, "oldObject", null) != null
  msg := sprintf("op=%v object=%v oldObject=%v", [op, has_obj, has_old])
}
'''


@pytest.mark.parametrize("backend", BACKENDS)
def test_augmented_unstructured_operation_and_delete(backend):
    """pkg/target/target_test.go:1218-1286 (TestAugmentedUnstructuredDeleteUsesOldObject / ...NonDeletePreservesObject): CREATE,
    UPDATE and a missing operation are preserved with the object in `object` and no oldObject; DELETE carries the object as
    oldObject, and HandleReview's setObjectOnDelete (target.go:269-287) then shows it as `object` too"""
    tmpl = {"apiVersion": "templates.gatekeeper.sh/v1", "kind": "ConstraintTemplate", "metadata": {"name": "k8sop"},
            "spec": {"crd": {"spec": {"names": {"kind": "K8sOp"}}}, "targets": [{"target": "admission.k8s.gatekeeper.sh", "rego": OPERATION_REGO}]}}
    con = {"apiVersion": "constraints.gatekeeper.sh/v1beta1", "kind": "K8sOp", "metadata": {"name": "x"}, "spec": {}}
    c, oc = load_both(backend, [tmpl], [con])
    thing = {"apiVersion": "some/v1", "kind": "Thing", "metadata": {"name": "foo"}}
    want = {"CREATE": "op=CREATE object=true oldObject=false", "UPDATE": "op=UPDATE object=true oldObject=false", "": "op= object=true oldObject=false",   # (AdmissionRequest.Operation has no omitempty: an empty string, not absent)
            "DELETE": "op=DELETE object=true oldObject=true"}
    rv = [D.AugmentedUnstructured(D.Unstructured(thing), None, "Original", op) for op in want]
    assert assert_parity(c, oc, rv) == 4
    got = c.ReviewBatch(rv, D.AUDIT_EP)
    assert [[r.msg for r in g] for g in got] == [[m] for m in want.values()]


@pytest.mark.parametrize("backend", BACKENDS)
def test_kernel_only_launches_leave_the_bitmaps_of_a_full_launch(backend, fixtures):
    """GK_EVAL_KERNEL_ONLY (the bench's back-to-back timing of the dominant kernel): launches without the totals kernel behind them write the same
    bitmaps as a full launch; the totals of a later full launch are untouched by them."""
    import numpy as np
    client = make_client(backend)
    for t in synth.psp_templates(fixtures):
        client.AddTemplate(t)
    for k in synth.audit_constraints():
        client.AddConstraint(k)
    drv = client.driver
    n = 700
    batch = synth.NativeBatch(drv.engine.lib, n, seed=synth.SEED, mixed=True, start=0, namespaces=synth.gen_namespaces())
    table = drv.engine.create_table_native(batch.reviews, n, keep_docs=False, resident=True)
    table.launch()
    full = table.eval(download=True, collect_only=True)
    viol, counts = np.array(full.viol, copy=True), np.array(full.counts, copy=True)
    assert counts.sum() > 100
    for _ in range(3):
        table.launch(kernel_only=True)
    only = table.eval(download=True, collect_only=True)
    assert np.array_equal(np.array(only.viol), viol)
    table.launch()
    again = table.eval(download=True, collect_only=True)
    assert np.array_equal(np.array(again.viol), viol) and np.array_equal(np.array(again.counts), counts)


@pytest.mark.parametrize("backend", BACKENDS)
def test_gator_verify_integer_vectors(backend, fixtures):
    """pkg/gator/verify/runner_integer_test.go TestRunner_Run_Integer (a reference-held vector pinned in round 6): K8sReplicaLimits in
    its three template flavours; 3 replicas: 0 violations, 100 replicas: exactly 1 -- on the device, and the message with the parameters
    object printed by %v equal to the oracle's; further replica counts around the range's ends (and a float, a string, none) by parity."""
    consts = fixtures["go_consts"]["pkg/gator/verify/runner_integer_test.go"]
    allow, deny = consts["objectIntegerAllowed"]["docs"][0], consts["objectIntegerDisallowed"]["docs"][0]
    import copy
    extra = []
    for i, rep in enumerate([2, 3, 50, 51, 3.0, 2.5, "3", None]):
        o = copy.deepcopy(allow)
        o["metadata"]["name"] = "d%d" % i
        if rep is None:
            del o["spec"]["replicas"]
        else:
            o["spec"]["replicas"] = rep
        extra.append(o)
    for tname, cname in (("templateV1Beta1Integer", "constraintV1Beta1Integer"), ("templateV1Beta1IntegerNonStructural", "constraintV1Beta1Integer"),
                         ("templateV1Integer", "constraintV1Integer")):
        c, oc = load_both(backend, [consts[tname]["docs"][0]], [consts[cname]["docs"][0]])
        rv = [D.AugmentedUnstructured(D.Unstructured(o), None, "Original") for o in [allow, deny] + extra]
        got = c.ReviewBatch(rv, D.GATOR_EP)
        assert len(got[0]) == 0 and len(got[1]) == 1
        assert got[1][0].msg == 'The provided number of replicas is not allowed for deployment: disallowed-deployment. Allowed ranges: {"ranges": [{"max_replicas": 50, "min_replicas": 3}]}'
        assert assert_parity(c, oc, rv, D.GATOR_EP) >= 5
