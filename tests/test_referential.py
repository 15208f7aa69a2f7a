"""Referential templates (data.inventory) on the device plan.  The synced objects (Client.AddData -> gk_data_put) are a
CONSTANT of the compiled formula: `other := data.inventory.namespace[ns][apiversion]["Ingress"][name]` unrolls into one
alternative per synced Ingress, `other.spec.rules[_].host == host` becomes a comparison of the review's host with that
object's hosts, `not identical(other, input.review)` a comparison of the review's name / namespace with its -- ordinary row
predicates.  The constraints of a referential template are compiled again whenever the synced objects change
(engine.cpp refresh_referential); an inventory that does not fit a plan makes every evaluation fail with GK_ERR_UNSUPPORTED
(closed: the cgo shim keeps such a template on the stock driver), never a guess.

Pinned by the reference: the message of test/gator/test/test.bats:222 for policies/default + manifests/referential-data
through the `gator test` harness; everything else product vs oracle.  K8sUniqueLabel (demo/basic, test/bats/test.bats:295-303
"unique labels test") compares sprintf("%v/%v", [group, version]) of the review with the synced objects' apiVersion: a formatted
string whose parts are known is compared with a constant piece by piece (pe.cpp fmt_equals).  K8sUniqueServiceSelector
(pkg/gator/fixtures/fixtures.go:414-471, test_test.go:134-170) compares a value COMPUTED from a whole sub-object by a closed
helper: a DEEP dictionary expression evaluated by the flattener (dexpr.hpp)."""
import pytest

import reference_tables as T
from conftest import gconst
from gatekeeper_amd import driver as D
from gatekeeper_amd import gator as G
from oracle import client as OC
from oracle import gator as OG
from oracle import target as OT
from parity_util import BACKENDS, make_client

GT = "test/gator/test/fixtures/"


def docs(fixtures, *paths):
    out = []
    for p in paths:
        out.extend(fixtures["yaml"][p]["docs"])
    return out


def ingress(name, ns, *hosts, api="networking.k8s.io/v1"):
    return {"apiVersion": api, "kind": "Ingress", "metadata": {"name": name, "namespace": ns}, "spec": {"rules": [{"host": h} for h in hosts]}}


def both(backend, fixtures):
    tmpl = docs(fixtures, GT + "policies/default/template_k8suniqueingresshost.yaml")[0]
    con = docs(fixtures, GT + "policies/default/constraint_k8suniqueingresshost.yaml")[0]
    c, oc = make_client(backend), OC.Client()
    c.AddTemplate(tmpl); oc.add_template(tmpl)
    c.AddConstraint(con); oc.add_constraint(con)
    return c, oc


def check(c, oc, objs, ep=D.GATOR_EP):
    got = c.ReviewBatch([D.AugmentedUnstructured(D.Unstructured(o), None, "Original") for o in objs], ep)
    n = 0
    for o, g in zip(objs, got):
        assert not isinstance(g, Exception), g
        want = sorted(r.msg for r in oc.review(OT.AugmentedUnstructured(OT.Unstructured(o), None, "Original"), ep))
        assert sorted(r.msg for r in g) == want, o["metadata"]["name"]
        n += len(want)
    return n


@pytest.mark.parametrize("backend", BACKENDS)
def test_bats_referential_data_through_the_gator_harness(backend, fixtures):
    """test.bats:211-223: `gator test -f policies/default -f manifests/referential-data` reports the ingress host conflict"""
    pol = sorted(p for p in fixtures["yaml"] if p.startswith(GT + "policies/default/"))
    ref = sorted(p for p in fixtures["yaml"] if p.startswith(GT + "manifests/referential-data/"))
    objs = docs(fixtures, *(pol + ref))
    c = make_client(backend)
    c.enforcement_points = (D.GATOR_EP,)
    got = G.test(objs, client=c)
    assert T.MSG_INGRESS in [g.msg for g in got]
    assert G.exit_code(got) == 1
    assert sorted((g.msg, g.violating_object["metadata"]["name"]) for g in got) == sorted((r.msg, o["metadata"]["name"]) for r, o in OG.gator_test(objs))


@pytest.mark.parametrize("backend", BACKENDS)
def test_the_compiled_constraint_follows_the_synced_objects(backend, fixtures):
    c, oc = both(backend, fixtures)
    a, b = ingress("a", "default", "x.example.com", api="extensions/v1beta1"), ingress("b", "default", "x.example.com", "y.example.com")
    other_ns, same_name = ingress("c", "prod", "y.example.com"), ingress("a", "prod", "z.example.com")
    not_ingress = {"apiVersion": "v1", "kind": "Service", "metadata": {"name": "svc", "namespace": "default"}, "spec": {"rules": [{"host": "x.example.com"}]}}
    fresh = ingress("new", "default", "x.example.com", "unique.example.com")
    everything = [a, b, other_ns, same_name, not_ingress, fresh]
    assert check(c, oc, everything) == 0                       # nothing synced: nothing to conflict with
    for o in (a, b):
        c.AddData(o); oc.add_data(o)
    assert check(c, oc, everything) == 4                       # a <-> b on x, c against b's y, the new one against both on x (ONE message: a set)
    for o in (other_ns, same_name, not_ingress):
        c.AddData(o); oc.add_data(o)
    assert check(c, oc, everything) == 5                       # + b's second host against c (synced now); a Service's hosts are no Ingress hosts
    c.RemoveData(a); oc.remove_data(a)
    assert check(c, oc, everything) == 5 - 1                   # b no longer conflicts on x, a (unsynced now) still does, with b
    b2 = ingress("b", "default", "q.example.com")
    c.AddData(b2); oc.add_data(b2)                             # a synced object CHANGES
    assert check(c, oc, [a, b2, other_ns, same_name, fresh]) == 0


@pytest.mark.parametrize("backend", BACKENDS)
def test_result_totals_of_a_referential_constraint(backend, fixtures):
    c, oc = both(backend, fixtures)
    objs = [ingress("i%d" % i, "ns%d" % (i % 3), "h%d.example.com" % (i // 2), "shared.example.com" if i % 5 == 0 else "own%d.example.com" % i) for i in range(24)]
    for o in objs:
        c.AddData(o); oc.add_data(o)
    want = sum(len(oc.review(OT.AugmentedUnstructured(OT.Unstructured(o), None, "Original"), OC.AUDIT_EP)) for o in objs)
    rep = c.AuditAggregate([D.AugmentedUnstructured(D.Unstructured(o), None, "Original") for o in objs], limit=3)
    assert not rep.errors and sum(v["total"] for v in rep.values()) == want
    assert want > sum(v["total_pairs"] for v in rep.values())          # two conflicting hosts on one Ingress: two results, one pair


def test_an_inventory_beyond_the_plan_fails_closed(fixtures):
    c, oc = both("hostemu", fixtures)
    objs = [ingress("i%d" % i, "default", "h%d.example.com" % i) for i in range(600)]
    for o in objs:
        c.AddData(o)
    with pytest.raises(D.UnsupportedError, match="does not compile against the synced inventory"):
        c.ReviewBatch([D.AugmentedUnstructured(D.Unstructured(objs[0]), None, "Original")], D.GATOR_EP)
    for o in objs[40:]:
        c.RemoveData(o)
    for o in objs[:40]:
        oc.add_data(o)
    dup = ingress("dup", "default", "h7.example.com")
    assert check(c, oc, [objs[0], dup]) == 1                   # the inventory fits again: the constraint serves again


@pytest.mark.parametrize("backend", BACKENDS)
def test_gator_referential_constraint_pinned_by_test_test_go(backend, fixtures):
    """pkg/gator/test/test_test.go:134-170 ("referential constraint with violation" / "without violation"): K8sUniqueServiceSelector
    (fixtures.go:414-471) joins on flatten_selector(obj) = concat(",", sort([concat(":", [k, v]) | v = obj.spec.selector[k]])) -- no
    formula over rows expresses that.  The helper is CLOSED (it reads nothing but its argument), so the partial evaluator hands
    the call to the flattener as a DEEP dictionary expression over the narrowest sub-document it looks at (object.spec.selector):
    the concrete evaluator runs the helper on every distinct selector once, the device tests a bit (dexpr.hpp)."""
    tmpl, con = gconst(fixtures, "TemplateReferential")[0], gconst(fixtures, "ConstraintReferential")[0]
    inv, deny, allow = (gconst(fixtures, k)[0] for k in ("ObjectReferentialInventory", "ObjectReferentialDeny", "ObjectReferentialAllow"))
    for objs, want in (([tmpl, con, inv, deny], ["same selector as service <gatekeeper-test-service-disallowed> in namespace <default>",
                                                 "same selector as service <gatekeeper-test-service-example> in namespace <default>"]),
                       ([tmpl, con, inv, allow], [])):
        c = make_client(backend)
        c.enforcement_points = (D.GATOR_EP,)
        got = G.test(objs, client=c)
        assert sorted(g.msg for g in got) == want
        assert sorted((g.msg, g.violating_object["metadata"]["name"]) for g in got) == sorted((r.msg, o["metadata"]["name"]) for r, o in OG.gator_test(objs))


@pytest.mark.parametrize("backend", BACKENDS)
def test_unique_service_selector_over_odd_selectors(backend, fixtures):
    """the deep expression against the oracle: key order, subsets and supersets, non-string values (skipped by the comprehension),
    separators inside keys and values ("a:b" / "c" flattens like "a" / "b:c"), selector absent / empty / not an object, a synced
    Service without a selector (flattens to "": equal to every review Service without string entries)"""
    tmpl, con = gconst(fixtures, "TemplateReferential")[0], gconst(fixtures, "ConstraintReferential")[0]
    c, oc = make_client(backend), OC.Client()
    c.AddTemplate(tmpl); oc.add_template(tmpl)

    def svc(name, ns, sel, kind="Service", api="v1"):
        spec = {"ports": [{"port": 80}]}
        if sel is not None:
            spec["selector"] = sel
        return {"apiVersion": api, "kind": kind, "metadata": {"name": name, "namespace": ns}, "spec": spec}
    inv = [svc("s1", "default", {"app": "a", "tier": "web"}), svc("s2", "other", {"k": "v"}), svc("s3", "default", {"a:b": "c"}), svc("s4", "default", {"x": "1,y:2"}),
           svc("none", "default", None), svc("dep", "default", {"k": "v"}, "Deployment", "apps/v1")]
    for o in inv:
        c.AddData(o); oc.add_data(o)
    c.AddConstraint(con); oc.add_constraint(con)
    objs = [svc("n1", "default", {"tier": "web", "app": "a"}), svc("s1", "default", {"app": "a", "tier": "web"}), svc("n2", "x", {"k": "v"}), svc("n3", "x", {"k": "w"}),
            svc("n4", "x", {}), svc("n5", "x", {"k": "v", "z": "1"}), svc("n6", "x", {"k": 5}), svc("n7", "x", {"k": "v", "n": 5}), svc("n8", "x", {"a": "b:c"}),
            svc("n9", "x", {"x": "1", "y": "2"}), svc("n10", "x", None), svc("n11", "x", ["k", "v"]), svc("n12", "x", "k:v"), svc("none", "default", None),
            svc("n13", "x", {"app": "a"}), svc("d2", "x", {"k": "v"}, "Deployment", "apps/v1"), svc("s2", "other", {"k": "v"}), svc("s2", "default", {"k": "v"})]
    assert check(c, oc, objs) >= 9
    assert check(c, oc, objs, D.AUDIT_EP) >= 9
    gone = inv[1]
    c.RemoveData(gone); oc.remove_data(gone)
    assert check(c, oc, objs) >= 6


def test_what_is_still_refused(fixtures):
    """an open helper (it reads input.parameters besides its argument) that sorts review data: no plan, no deep expression"""
    rego = """package k
norm(obj) = out { out := concat(",", sort([x | x = obj.spec.names[_]; x != input.parameters.skip])) }
violation[{"msg": "m"}] { norm(input.review.object) == "a,b" }
"""
    tmpl = {"apiVersion": "templates.gatekeeper.sh/v1", "kind": "ConstraintTemplate", "metadata": {"name": "k8sopen"},
            "spec": {"crd": {"spec": {"names": {"kind": "K8sOpen"}}}, "targets": [{"target": "admission.k8s.gatekeeper.sh", "rego": rego}]}}
    c = make_client("hostemu")
    c.AddTemplate(tmpl)
    with pytest.raises(D.UnsupportedError, match="sort applied to review data"):
        c.AddConstraint({"apiVersion": "constraints.gatekeeper.sh/v1beta1", "kind": "K8sOpen", "metadata": {"name": "x"}, "spec": {"parameters": {"skip": "z"}}})


# ---- K8sUniqueLabel: demo/basic + the bats "unique labels test" (test/bats/test.bats:295-303: with no_dupe_cm synced, applying
# bad/no_dupe_cm_2.yaml is denied)
def unique_label(backend, fixtures, constraint_path):
    tmpl = docs(fixtures, "test/bats/tests/templates/k8suniquelabel_template.yaml")[0]
    con = docs(fixtures, constraint_path)[0]
    c, oc = make_client(backend), OC.Client()
    c.AddTemplate(tmpl); oc.add_template(tmpl)
    c.AddConstraint(con); oc.add_constraint(con)
    return c, oc


@pytest.mark.parametrize("backend", BACKENDS)
def test_bats_unique_labels(backend, fixtures):
    c, oc = unique_label(backend, fixtures, "test/bats/tests/constraints/all_cm_gatekeeper_label_unique.yaml")
    good = docs(fixtures, "test/bats/tests/good/no_dupe_cm.yaml")[0]
    bad = docs(fixtures, "test/bats/tests/bad/no_dupe_cm_2.yaml")[0]
    assert check(c, oc, [good, bad]) == 0                       # nothing synced yet: no duplicate
    c.AddData(good); oc.add_data(good)
    got = c.ReviewBatch([D.AugmentedUnstructured(D.Unstructured(o), None, "Original") for o in (bad, good)], D.WEBHOOK_EP)
    assert [r.msg for r in got[0]] == ["label gatekeeper has duplicate value not_duplicated"] and got[0][0].enforcement_action == "deny"
    assert got[1] == []                                          # the synced object itself is not its own duplicate
    assert check(c, oc, [good, bad], D.WEBHOOK_EP) == 1
    c.RemoveData(good); oc.remove_data(good)
    assert check(c, oc, [good, bad]) == 0


@pytest.mark.parametrize("backend", BACKENDS)
def test_unique_label_over_cluster_and_namespaced_objects(backend, fixtures):
    """identical_cluster / identical_namespace: same name in another namespace, same name and namespace under another kind or
    group/version ("apps/v1" against make_apiversion's sprintf), label absent, label value shared by a cluster-scoped object"""
    c, oc = unique_label(backend, fixtures, "demo/basic/constraints/all_ns_gatekeeper_label_unique.yaml")
    con_all = {"apiVersion": "constraints.gatekeeper.sh/v1beta1", "kind": "K8sUniqueLabel", "metadata": {"name": "everything"}, "spec": {"parameters": {"label": "team"}}}
    c.AddConstraint(con_all); oc.add_constraint(con_all)

    def ns(name, labels): return {"apiVersion": "v1", "kind": "Namespace", "metadata": {"name": name, "labels": labels}}
    def cm(name, nsn, labels, kind="ConfigMap", api="v1"): return {"apiVersion": api, "kind": kind, "metadata": {"name": name, "namespace": nsn, "labels": labels}}
    inv = [ns("a", {"gatekeeper": "x", "team": "red"}), ns("b", {"gatekeeper": "y"}), ns("c", {}), cm("m1", "a", {"team": "blue"}), cm("m2", "b", {"other": "x"}),
           cm("d1", "a", {"team": "green"}, "Deployment", "apps/v1"), cm("d1", "a", {"team": "teal"}, "Deployment", "extensions/v1beta1")]
    for o in inv:
        c.AddData(o); oc.add_data(o)
    objs = [ns("d", {"gatekeeper": "x"}), ns("a", {"gatekeeper": "x", "team": "red"}), ns("e", {"gatekeeper": "z", "team": "blue"}), ns("f", {}),
            cm("m1", "a", {"team": "blue"}), cm("m3", "a", {"team": "blue"}), cm("m1", "b", {"team": "blue"}), cm("zz", "b", {"team": "red"}),
            cm("d1", "a", {"team": "green"}, "Deployment", "apps/v1"), cm("d1", "a", {"team": "green"}, "Deployment", "apps/v1beta2"), cm("d1", "a", {"team": "green"}, "StatefulSet", "apps/v1"),
            cm("d1", "a", {"team": "teal"}, "Deployment", "extensions/v1beta1"), cm("d1", "a", {"team": "teal"}, "Deployment", "extensions/v1"),
            cm("d1", "a", {"team": "teal"}, "Deployment", "v1beta1"), cm("q", "a", {"team": 7}), cm("q", "a", {"team": ""}), cm("q", "a", None)]
    assert check(c, oc, objs) >= 9
    assert check(c, oc, objs, D.AUDIT_EP) >= 9


FMT = """package k
violation[{"msg": msg}] {
  want := sprintf("%v/%v:%v", [input.review.kind.group, input.review.kind.kind, input.review.name])
  want == input.parameters.ids[_]
  msg := sprintf("listed %v", [want])
}
violation[{"msg": msg}] {
  sprintf("%s-%s", [input.review.namespace, input.review.name]) != input.parameters.not_this
  input.review.kind.kind == "Secret"
  msg := "another secret"
}
violation[{"msg": msg}] {
  sprintf("%v%v", [input.review.name, input.review.namespace]) == "abab"
  msg := "two adjacent verbs"
}
"""


@pytest.mark.parametrize("backend", BACKENDS)
def test_formatted_review_strings_against_constants(backend):
    """sprintf over review leaves that are strings by construction (review.kind.*, name, namespace) compared with constants: every way
    of cutting the constant (separators inside the pieces, adjacent verbs, empty pieces), != and undefined operands"""
    tmpl = {"apiVersion": "templates.gatekeeper.sh/v1", "kind": "ConstraintTemplate", "metadata": {"name": "k8sfmt"},
            "spec": {"crd": {"spec": {"names": {"kind": "K8sFmt"}}}, "targets": [{"target": "admission.k8s.gatekeeper.sh", "rego": FMT}]}}
    con = {"apiVersion": "constraints.gatekeeper.sh/v1beta1", "kind": "K8sFmt", "metadata": {"name": "x"},
           "spec": {"parameters": {"ids": ["apps/Deployment:web", "/Pod:a/b:c", "/Pod:", "x/y/Z:n", 5, "/ConfigMap:7"], "not_this": "ns1-s1"}}}
    c, oc = make_client(backend), OC.Client()
    c.AddTemplate(tmpl); oc.add_template(tmpl)
    c.AddConstraint(con); oc.add_constraint(con)

    def o(api, kind, name, ns=None):
        md = {"name": name}
        if ns is not None:
            md["namespace"] = ns
        return {"apiVersion": api, "kind": kind, "metadata": md}
    objs = [o("apps/v1", "Deployment", "web", "d"), o("apps/v1", "Deployment", "web2", "d"), o("v1", "Pod", "a/b:c", "d"), o("v1", "Pod", "", "d"), o("x/y/v1", "Z", "n", "d"),
            o("v1", "ConfigMap", "7", "d"), o("v1", "Secret", "s1", "ns1"), o("v1", "Secret", "s2", "ns1"), o("v1", "Secret", "s1-x", "ns1"), o("v1", "Secret", "clusterwide"),
            o("v1", "Pod", "ab", "ab"), o("v1", "Pod", "a", "bab"), o("v1", "Pod", "aba", "b"), o("v1", "Pod", "abab", ""), o("v1", "Pod", "ba", "ba")]
    assert check(c, oc, objs) >= 8


@pytest.mark.parametrize("backend", BACKENDS)
def test_replacing_one_referential_template_leaves_no_other_kind_on_a_stale_inventory(backend, fixtures):
    """round-3 advisor finding (fail-open): two referential kinds; a data change with no review in between, then AddTemplate of ONE
    kind.  gk_template_add recompiled that kind against the new inventory and marked the inventory compiled -- the other kind's
    constraint kept the older snapshot and missed the violation until the next data change."""
    tmpl_a = docs(fixtures, GT + "policies/default/template_k8suniqueingresshost.yaml")[0]
    con_a = docs(fixtures, GT + "policies/default/constraint_k8suniqueingresshost.yaml")[0]
    import copy
    tmpl_b = copy.deepcopy(tmpl_a)
    tmpl_b["metadata"]["name"] = "k8suniqueingresshostb"
    tmpl_b["spec"]["crd"]["spec"]["names"]["kind"] = "K8sUniqueIngressHostB"
    for t in tmpl_b["spec"]["targets"]:
        t["rego"] = t["rego"].replace("package k8suniqueingresshost", "package k8suniqueingresshostb")
    con_b = copy.deepcopy(con_a)
    con_b["kind"] = "K8sUniqueIngressHostB"
    con_b["metadata"]["name"] = "unique-ingress-host-b"
    c, oc = make_client(backend), OC.Client()
    for t in (tmpl_a, tmpl_b):
        c.AddTemplate(t); oc.add_template(t)
    for k in (con_a, con_b):
        c.AddConstraint(k); oc.add_constraint(k)
    a, b = ingress("a", "default", "a.example.com"), ingress("b", "default", "b.example.com")
    probe = ingress("new", "default", "b.example.com")
    c.AddData(a); oc.add_data(a)
    assert check(c, oc, [probe]) == 0
    c.AddData(b); oc.add_data(b)                               # no review in between
    edited = copy.deepcopy(tmpl_b)
    for t in edited["spec"]["targets"]:
        t["rego"] = t["rego"].replace("ingress host conflicts", "INGRESS host conflicts")
    assert edited != tmpl_b
    c.AddTemplate(edited); oc.add_template(edited)
    assert check(c, oc, [probe]) == 2                          # BOTH kinds see b
