"""Pins the oracle's Match layer against the reference's own table tests."""
import pytest

import reference_tables as T
from oracle import audit, match
from oracle import target as tg
from oracle.client import AUDIT_EP, Client


@pytest.mark.parametrize("w,cand,want", T.WILDCARD_MATCHES)
def test_wildcard_matches(w, cand, want):
    assert match.wildcard_matches(w, cand) is want


@pytest.mark.parametrize("w,cand,want", T.WILDCARD_GENERATE_NAME)
def test_wildcard_generate_name(w, cand, want):
    assert match.wildcard_matches_generate_name(w, cand) is want


@pytest.mark.parametrize("case", T.MATCH_CASES, ids=[c[0] for c in T.MATCH_CASES])
def test_match_table(case):
    _, obj, mt, ns, source, want, want_err = case
    if want_err:
        with pytest.raises(match.MatchError) as ei:
            match.matches(mt, obj, ns, source)
        assert str(ei.value).startswith(match.ERR_MATCH)
    else:
        assert match.matches(mt, obj, ns, source) is want


@pytest.mark.parametrize("case", T.NAMES_MATCH_CASES, ids=[c[0] for c in T.NAMES_MATCH_CASES])
def test_names_match(case):
    _, name, obj, want = case
    assert match.names_match({"name": name}, obj, None, "") is want


DENYALL = None


def _denyall(fixtures):
    return fixtures["go_consts"]["pkg/target/target_integration_test.go"]["testTemplate"]["docs"][0]


@pytest.mark.parametrize("case", T.ENFORCEMENT_CASES, ids=[c[0] for c in T.ENFORCEMENT_CASES])
def test_constraint_enforcement(case, fixtures):
    """pkg/target/target_integration_test.go:163-527: allowed <=> zero results, in all three review shapes."""
    _, obj, ns, mt, allowed = case
    c = Client(enforcement_points=(AUDIT_EP,))
    c.add_template(_denyall(fixtures))
    cons = {"apiVersion": "constraints.gatekeeper.sh/v1beta1", "kind": "DenyAll", "metadata": {"name": "my-constraint"}}
    if mt is not None:
        cons["spec"] = {"match": mt}
    c.add_constraint(cons)
    g, v, k = match.obj_gvk(obj)
    kind = {"group": g, "version": v, "kind": k}
    req = tg.AdmissionRequest({"kind": kind, "object": obj})
    req2 = tg.AdmissionRequest({"kind": kind, "oldObject": obj})
    if ns is not None:
        req["namespace"] = ns["metadata"]["name"]
        req2["namespace"] = ns["metadata"]["name"]
    for review in (tg.AugmentedReview(req, ns), tg.AugmentedReview(req2, ns), tg.AugmentedUnstructured(tg.Unstructured(obj), ns)):
        res = c.review(review, AUDIT_EP)
        assert (len(res) == 0) is allowed
        if not allowed:
            assert res[0].msg == "denyall constraint installed"


def test_delete_handling():
    """pkg/target/target.go:269-287 / target_test.go:1154-1216"""
    obj = {"apiVersion": "v1", "kind": "Pod", "metadata": {"name": "p"}}
    with pytest.raises(tg.ReviewError) as ei:
        tg.handle_review(tg.AdmissionRequest({"operation": "DELETE", "object": obj}))
    assert str(ei.value) == tg.ERR_OLD_OBJECT_IS_NIL
    handled, r = tg.handle_review(tg.AdmissionRequest({"operation": "DELETE", "oldObject": obj}))
    assert handled and r.request["object"] == obj
    handled, r = tg.handle_review(tg.AugmentedUnstructured(tg.Unstructured(obj), operation="DELETE"))
    assert handled and r.request["object"] == obj and r.request["oldObject"] == obj
    assert tg.handle_review("not a review") == (False, None)


def test_process_data():
    """pkg/target/target.go:40-66 / target_test.go:401"""
    pod = tg.Unstructured({"apiVersion": "v1", "kind": "Pod", "metadata": {"name": "p", "namespace": "ns"}})
    assert tg.process_data(pod)[1] == ["namespace", "ns", "v1", "Pod", "p"]
    dep = tg.Unstructured({"apiVersion": "apps/v1", "kind": "Deployment", "metadata": {"name": "d"}})
    assert tg.process_data(dep)[1] == ["cluster", "apps/v1", "Deployment", "d"]
    with pytest.raises(tg.ReviewError):
        tg.process_data(tg.Unstructured({"kind": "Pod", "metadata": {"name": "x"}}))


def test_ns_cache_lookup():
    """matcher.go:37-39: namespaceSelector falls back to the Namespace cached through AddData."""
    c = Client()
    c.add_data({"apiVersion": "v1", "kind": "Namespace", "metadata": {"name": "foo", "labels": {"bar": "qux"}}})
    mt = tg.Matcher({"namespaceSelector": {"matchLabels": {"bar": "qux"}}}, c.cache)
    obj = {"apiVersion": "v1", "kind": "Pod", "metadata": {"name": "p", "namespace": "foo"}}
    rv = tg.GkReview(tg.AdmissionRequest({"object": obj, "namespace": "foo"}))
    assert mt.match_review(rv) is True
    rv2 = tg.GkReview(tg.AdmissionRequest({"object": obj, "namespace": "other"}))
    with pytest.raises(tg.ReviewError) as ei:
        mt.match_review(rv2)
    assert str(ei.value) == ("error matching the requested object: p :failed to run Match criteria: "
                             "namespace selector for namespace-scoped object but missing Namespace")


@pytest.mark.parametrize("s,size,want", T.TRUNCATE_CASES)
def test_truncate(s, size, want):
    assert audit.truncate_string(s, size) == want


def test_limit_queue():
    items = [dict(name="", message="", enforcementAction="", **i) for i in T.SVQ_ITEMS]
    desc = sorted(range(3), key=lambda i: audit.sv_key(items[i]), reverse=True)
    assert desc == T.SVQ_POP_ORDER
    q = audit.LimitQueue(2)
    for it in items:
        q.push(it)
    assert sorted(items.index(x) for x in q.items) == T.LIMITQ2_REMAINING


def test_process_validation_results_a12():
    """Row a12: deny / warn message lists (pkg/webhook/policy.go:265-399), oracle and product host mirror against the
    reference's own table (policy_test.go:1303-1339, :1395-1536) and the "[<name>] <msg>" format (:390,394)."""
    from golden.reference_tables import PROCESS_RESULTS_CASES
    from oracle import client as OC
    from oracle import webhook as OW
    from gatekeeper_amd import driver as D
    for name, rows, n_deny, n_warn in PROCESS_RESULTS_CASES:
        for mod, proc in ((OC, OW.process_validation_results), (D, D.process_validation_results)):
            res = []
            for r in rows:
                if r is None:
                    res.append(None)
                    continue
                msg, cname, ea, scoped = r
                c = {"kind": "Foo", "metadata": {"name": cname}, "spec": {"enforcementAction": ea}} if cname else None
                res.append(mod.Result(msg, c, {}, ea, scoped))
            deny, warn = proc(res)
            assert (len(deny), len(warn)) == (n_deny, n_warn), (name, mod.__name__)
    deny, warn = D.process_validation_results([D.Result("m1", {"metadata": {"name": "c1"}}, {}, "deny", None),
                                               D.Result("m2", {"metadata": {"name": "c2"}}, {}, "scoped", ["warn", "bogus"])])
    assert deny == ["[c1] m1"] and warn == ["[c2] m2"]


def test_validate_constraint_table(fixtures):
    """pkg/target/target_test.go:42-399 (TestValidateConstraint, rows extracted by make_golden.py): the target handler's
    check run at AddConstraint, in the oracle and in the host mirror."""
    from oracle import target as OT
    from gatekeeper_amd import driver as D
    rows = fixtures["validate_constraint_cases"]
    assert len(rows) == 11 and sum(r["error_expected"] for r in rows) == 6
    for r in rows:
        for fn, exc in ((OT.validate_constraint, OT.ReviewError), (D.validate_constraint, D.ClientError)):
            if r["error_expected"]:
                with pytest.raises(exc):
                    fn(r["constraint"])
            else:
                fn(r["constraint"])
    # demo/basic/bad/bad_constraint_labelselector.yaml: the demo's deliberately invalid constraint (operator In, no values)
    bad = fixtures["yaml"]["demo/basic/bad/bad_constraint_labelselector.yaml"]["docs"][0]
    with pytest.raises(OT.ReviewError):
        OT.validate_constraint(bad)
    with pytest.raises(D.ClientError):
        D.validate_constraint(bad)


def test_namespace_cache_table():
    """pkg/target/target_test.go:983-1152 (TestNamespaceCache): add / remove / lookup, non-Namespace objects are ignored,
    a Namespace that does not convert into the typed object is refused (ErrCachingType)."""
    def ns(name, labels):
        return {"apiVersion": "v1", "kind": "Namespace", "metadata": {"name": name, "labels": labels}}
    foo_constraint = {"apiVersion": "constraints.gatekeeper.sh/v1beta1", "kind": "Foo", "metadata": {"name": "c"}, "spec": {"match": {}}}
    cases = [
        ([], [], [("my-ns1", False)], False),
        ([ns("my-ns1", {"ns1": "label"})], [], [("my-ns1", True), ("my-ns2", False)], False),
        ([ns("my-ns1", {"ns1": "label"}), ns("my-ns2", {"ns2": "label"})], [], [("my-ns1", True), ("my-ns2", True)], False),
        ([{"apiVersion": "v1", "kind": "Namespace", "spec": 3.0}], [], [], True),
        ([foo_constraint, ns("my-ns2", {"ns2": "label"})], [], [("my-ns2", True)], False),
        ([ns("my-ns1", {"ns1": "label"}), ns("my-ns2", {"ns2": "label"})], ["my-ns1"], [("my-ns1", False), ("my-ns2", True)], False),
    ]
    for adds, removes, checks, want_err in cases:
        cache = tg.NsCache()
        errs = 0
        for o in adds:
            _, key, _ = tg.process_data(tg.Unstructured(o))
            try:
                cache.add(key, o)
            except tg.ReviewError:
                errs += 1
        assert (errs > 0) is want_err
        for name in removes:
            _, key, _ = tg.process_data(tg.Unstructured({"apiVersion": "v1", "kind": "Namespace", "metadata": {"name": name}}))
            cache.remove(key)
        assert len(cache.cache) == sum(1 for _, e in checks if e)
        for name, exists in checks:
            got = cache.get_namespace(name)
            assert (got is not None) is exists
            if exists:
                assert got["metadata"]["name"] == name and got["metadata"]["labels"]


def test_enforcement_action_tables():
    """pkg/util/enforcement_action_test.go:113-165 (GetEnforcementAction) and :235-385 (ScopedActionForEP), oracle and host
    mirror."""
    from oracle import client as OC
    from gatekeeper_amd import driver as D
    AUDIT, WEBHOOK = "audit.gatekeeper.sh", "validation.gatekeeper.sh"

    def sea(*pairs):
        return {"spec": {"scopedEnforcementActions": [{"action": a, "enforcementPoints": [{"name": n} for n in eps]} for a, eps in pairs]}}
    for mod in (OC, D):
        assert mod.get_enforcement_action({}) == "deny"
        with pytest.raises(mod.ClientError):
            mod.get_enforcement_action({"spec": []})
        assert mod.get_enforcement_action({"spec": {"enforcementAction": "notsupported"}}) == "unrecognized"
        assert mod.get_enforcement_action({"spec": {"enforcementAction": "dryrun"}}) == "dryrun"
        f = mod.scoped_actions_for_ep
        assert f(AUDIT, sea(("deny", [AUDIT]), ("warn", [WEBHOOK]))) == ["deny"]
        assert f(WEBHOOK, sea(("deny", [AUDIT, WEBHOOK]), ("warn", [WEBHOOK]))) == ["deny", "warn"]
        assert f(AUDIT, sea(("deny", [WEBHOOK]), ("warn", [WEBHOOK]))) == []
        assert f(AUDIT, sea(("deny", ["*"]), ("warn", [WEBHOOK]))) == ["deny"]
        assert f(AUDIT, {"spec": {}}) == []
        with pytest.raises(mod.ClientError):
            f(AUDIT, {"spec": {"scopedEnforcementActions": "invalid"}})


def test_to_matcher_rows():
    """pkg/target/target_test.go:562-655 (TestToMatcher): no match fields -> a matcher of everything; fooConstraint's match fields ->
    a matcher; spec.match of the wrong type (3.0) and a match FIELD of the wrong type (kinds: 3.0) -> ErrCreatingMatcher.  Oracle and
    product mirror (driver.check_matcher) refuse the same constraints, validated or not (ToMatcher is not ValidateConstraint)."""
    from gatekeeper_amd import driver as D
    from oracle import client as OC
    foo = {"kinds": [{"apiGroups": ["some"], "kinds": ["Thing"]}], "scope": "Namespaced", "namespaces": ["my-ns"],
           "labelSelector": {"matchLabels": {"obj": "label"}}, "namespaceSelector": {"matchLabels": {"ns": "label"}}, "source": "All"}
    rows = [("no match fields", None, True), ("match fields", foo, True), ("invalid Match type", 3.0, False), ("invalid Match field type", {"kinds": 3.0}, False),
            # the same converter refuses every other wrongly typed field; unknown fields are ignored, null is the zero value
            ("namespaces not a list", {"namespaces": "my-ns"}, False), ("kind entry not a map", {"kinds": ["Thing"]}, False), ("apiGroups of numbers", {"kinds": [{"apiGroups": [1]}]}, False),
            ("matchLabels value not a string", {"labelSelector": {"matchLabels": {"a": 1}}}, False), ("expression values not a list", {"namespaceSelector": {"matchExpressions": [{"key": "a", "operator": "In", "values": "x"}]}}, False),
            ("scope a number", {"scope": 1}, False), ("unknown field", {"nonesuch": 3.0}, True), ("null fields", {"kinds": None, "labelSelector": None, "name": None}, True)]
    tmpl = {"apiVersion": "templates.gatekeeper.sh/v1", "kind": "ConstraintTemplate", "metadata": {"name": "k8sx"},
            "spec": {"crd": {"spec": {"names": {"kind": "K8sX"}}}, "targets": [{"target": "admission.k8s.gatekeeper.sh", "rego": 'package k\nviolation[{"msg": "m"}] { true }\n'}]}}
    for name, match, ok in rows:
        con = {"apiVersion": "constraints.gatekeeper.sh/v1beta1", "kind": "K8sX", "metadata": {"name": "x"}, "spec": ({"match": match} if match is not None else {})}
        if ok:
            tg.to_matcher(con, tg.NsCache())
            D.check_matcher(con)
        else:
            with pytest.raises(tg.ReviewError, match=tg.ERR_CREATING_MATCHER):
                tg.to_matcher(con, tg.NsCache())
            with pytest.raises(D.ClientError, match="unable to create matcher"):
                D.check_matcher(con)
            oc = OC.Client()
            oc.add_template(tmpl)
            with pytest.raises(OC.ClientError):
                oc.add_constraint(con, validate=False)
    # unstructured.NestedMap's accessor error on the way down: a spec that is no map is refused by ToMatcher and by ValidateConstraint;
    # a null spec is "not found"
    for spec, ok in (("text", False), ([1], False), (3.0, False), (None, True)):
        con = {"apiVersion": "constraints.gatekeeper.sh/v1beta1", "kind": "K8sX", "metadata": {"name": "x"}, "spec": spec}
        if ok:
            tg.to_matcher(con, tg.NsCache())
            tg.validate_constraint(con)
            D.check_matcher(con)
            D.validate_constraint(con)
            continue
        with pytest.raises(tg.ReviewError, match=tg.ERR_CREATING_MATCHER):
            tg.to_matcher(con, tg.NsCache())
        with pytest.raises(tg.ReviewError, match="accessor error"):
            tg.validate_constraint(con)
        with pytest.raises(D.ClientError, match="unable to create matcher"):
            D.check_matcher(con)
        with pytest.raises(D.ClientError):
            D.validate_constraint(con)


def test_constraint_validation_rows():
    """pkg/webhook/policy_test.go:686-802 (TestConstraintValidation) with the constraints of :49-139: a label / namespace selector
    whose `In` expression has no values is refused, the same with values is accepted, by the oracle's and the product mirror's
    AddConstraint alike.  (The sixth pair, enforcementAction "test", is refused by the webhook's validation of Constraint RESOURCES --
    policy.go validateGatekeeperResources, control plane -- not by the client's AddConstraint: out of this path.)"""
    from gatekeeper_amd import driver as D
    from oracle import client as OC
    tmpl = {"apiVersion": "templates.gatekeeper.sh/v1", "kind": "ConstraintTemplate", "metadata": {"name": "k8sgoodrego"},
            "spec": {"crd": {"spec": {"names": {"kind": "K8sGoodRego"}}}, "targets": [{"target": "admission.k8s.gatekeeper.sh", "rego": 'package k\nviolation[{"msg": "m"}] { true }\n'}]}}

    def con(name, match, ea=None):
        spec = {"match": match}
        if ea:
            spec["enforcementAction"] = ea
        return {"apiVersion": "constraints.gatekeeper.sh/v1beta1", "kind": "K8sGoodRego", "metadata": {"name": name}, "spec": spec}
    ns_k, pod_k = [{"apiGroups": [""], "kinds": ["Namespace"]}], [{"apiGroups": [""], "kinds": ["Pod"]}]
    good, bad = [{"operator": "In", "key": "something", "values": ["anything"]}], [{"operator": "In", "key": "something"}]
    rows = [(con("good-labelselector", {"kinds": ns_k, "labelSelector": {"matchExpressions": good}}), False),
            (con("bad-labelselector", {"kinds": ns_k, "labelSelector": {"matchExpressions": bad}}), True),
            (con("good-namespaceselector", {"kinds": pod_k, "namespaceSelector": {"matchExpressions": good}}), False),
            (con("bad-namespaceselector", {"kinds": pod_k, "namespaceSelector": {"matchExpressions": bad}}), True),
            (con("good-enforcementaction", {"kinds": pod_k}, "dryrun"), False)]
    for k, want_err in rows:
        oc = OC.Client()
        oc.add_template(tmpl)
        pc = D.Client(D.Driver(hostemu=True))
        pc.AddTemplate(tmpl)
        if want_err:
            with pytest.raises(OC.ClientError):
                oc.add_constraint(k)
            with pytest.raises(D.ClientError):
                pc.AddConstraint(k)
        else:
            oc.add_constraint(k)
            pc.AddConstraint(k)
