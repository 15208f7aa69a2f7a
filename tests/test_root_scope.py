"""Review values compared with EACH OTHER outside any iteration -- the shape of the "immutable field" policies
(`input.review.object.spec.serviceAccountName != input.review.oldObject.spec.serviceAccountName`, the public library's
noupdateserviceaccount), `name == namespace`, and an array element against a value outside its array.  The plan keeps such
values in a one-element ROOT scope (plan.hpp GK_LEVEL_ROOT: a stored value marks the element; the comparison runs inside
the scope's single-trip loop), so they use the same value slots and the same val_eq as joins between array elements:
short and long strings, numbers (3 == 3.0, 3 != "3"), a side that is missing (undefined, not "different")."""
import pytest

from gatekeeper_amd import driver as D
from parity_util import BACKENDS, assert_parity, load_both


def tmpl(kind, rego):
    return {"apiVersion": "templates.gatekeeper.sh/v1", "kind": "ConstraintTemplate", "metadata": {"name": kind.lower()},
            "spec": {"crd": {"spec": {"names": {"kind": kind}}}, "targets": [{"target": "admission.k8s.gatekeeper.sh", "rego": rego}]}}
T = {
 "K8sNoUpdateSA": '''package noupdateserviceaccount
violation[{"msg": msg}] {
  input.review.operation == "UPDATE"
  new := input.review.object.spec.serviceAccountName
  old := input.review.oldObject.spec.serviceAccountName
  new != old
  msg := sprintf("cannot update serviceAccountName from %v to %v", [old, new])
}
''',
 "K8sNameIsNamespace": '''package k
violation[{"msg": msg}] {
  input.review.object.metadata.name == input.review.object.metadata.namespace
  msg := sprintf("name %v equals its namespace", [input.review.object.metadata.name])
}
''',
 "K8sContainerNamedLikePod": '''package k
violation[{"msg": msg}] {
  c := input.review.object.spec.containers[_]
  c.name == input.review.object.metadata.name
  msg := sprintf("container %v is named like the pod", [c.name])
}
''',
 "K8sImmutableReplicasAndImage": '''package k
violation[{"msg": msg}] {
  input.review.oldObject.spec.replicas != input.review.object.spec.replicas
  msg := "replicas changed"
}
violation[{"msg": msg}] {
  not input.review.oldObject.spec.paused == input.review.object.spec.paused
  input.review.oldObject
  msg := "paused differs or is missing"
}
''',
}
def pod(name, ns, sa=None, containers=("a",), **spec):
    s = {"containers": [{"name": c, "image": "x"} for c in containers]}
    if sa is not None: s["serviceAccountName"] = sa
    s.update(spec)
    return {"apiVersion": "v1", "kind": "Pod", "metadata": {"name": name, "namespace": ns}, "spec": s}
def req(op, obj, old=None):
    r = {"uid": "u", "kind": {"group": "", "version": "v1", "kind": "Pod"}, "operation": op, "name": obj["metadata"]["name"], "namespace": obj["metadata"]["namespace"],
         "userInfo": {"username": "bob"}, "object": obj}
    if old is not None: r["oldObject"] = old
    return D.AugmentedReview(D.AdmissionRequest(r), None, "Original")
reviews = [
 req("UPDATE", pod("p", "d", "sa2"), pod("p", "d", "sa1")),
 req("UPDATE", pod("p", "d", "sa1"), pod("p", "d", "sa1")),
 req("UPDATE", pod("p", "d", "a-service-account-with-a-long-name-2"), pod("p", "d", "a-service-account-with-a-long-name-1")),
 req("UPDATE", pod("p", "d", "a-service-account-with-a-long-name-1"), pod("p", "d", "a-service-account-with-a-long-name-1")),
 req("UPDATE", pod("p", "d", "sa1"), pod("p", "d")),
 req("UPDATE", pod("p", "d"), pod("p", "d", "sa1")),
 req("CREATE", pod("p", "d", "sa2")),
 req("UPDATE", pod("same", "same", "x", containers=("same", "b")), pod("same", "same", "x")),
 req("UPDATE", pod("q", "d", "x", containers=("q",), replicas=3, paused=True), pod("q", "d", "x", replicas=2, paused=True)),
 req("UPDATE", pod("q", "d", "x", replicas=3, paused=False), pod("q", "d", "x", replicas=3.0, paused=True)),
 req("UPDATE", pod("q", "d", "x", replicas=3), pod("q", "d", "x", replicas="3")),
 D.AugmentedUnstructured(D.Unstructured(pod("plain", "plain", "x", containers=("plain",))), None, "Original"),
]


@pytest.mark.parametrize("backend", BACKENDS)
def test_values_compared_outside_iterations(backend):
    c, oc = load_both(backend, [tmpl(k, r) for k, r in T.items()],
                      [{"apiVersion": "constraints.gatekeeper.sh/v1beta1", "kind": k, "metadata": {"name": "x"}, "spec": {}} for k in T])
    assert assert_parity(c, oc, reviews, D.GATOR_EP) == 20
    got = c.ReviewBatch(reviews, D.GATOR_EP)
    sa = [[r.msg for r in g if r.constraint["kind"] == "K8sNoUpdateSA"] for g in got]
    assert sa[0] == ["cannot update serviceAccountName from sa1 to sa2"] and sa[1] == [] and len(sa[2]) == 1 and sa[3] == []
    assert sa[4] == sa[5] == sa[6] == []          # a missing side is undefined, CREATE is not UPDATE
    rep = [any(r.msg == "replicas changed" for r in g) for g in got]
    assert rep[8] and not rep[9] and rep[10]      # 2 -> 3; 3.0 == 3; "3" != 3


@pytest.mark.parametrize("backend", BACKENDS)
def test_composite_operands_are_refused_not_guessed(backend):
    """Rego's `==` between two review values is DEEP equality.  The plan compares type and payload: exact for scalars and for
    empty containers; for a non-empty container it would have to guess -- no kernel answers such a review whatever the other side
    holds (the row code shared with the device decides that); since round 5 the engine's host evaluator does, with Rego's answer:
    equal selectors are a violation, different ones are not."""
    rego = '''package k
violation[{"msg": msg}] {
  input.review.object.spec.selector == input.review.oldObject.spec.selector
  msg := "selector unchanged"
}
'''
    c, oc = load_both(backend, [tmpl("K8sSel", rego)], [{"apiVersion": "constraints.gatekeeper.sh/v1beta1", "kind": "K8sSel", "metadata": {"name": "x"}, "spec": {}}])

    def upd(new, old):
        def svc(sel):
            return {"apiVersion": "v1", "kind": "Service", "metadata": {"name": "s", "namespace": "d"}, "spec": {"selector": sel}}
        return D.AugmentedReview(D.AdmissionRequest({"uid": "u", "kind": {"group": "", "version": "v1", "kind": "Service"}, "operation": "UPDATE", "name": "s",
                                                     "namespace": "d", "object": svc(new), "oldObject": svc(old)}), None, "Original")
    cases = [({"app": "a"}, {"app": "a"}), ({"app": "a"}, {"app": "b"}), ({"app": "a"}, "a"), ({}, {}), ({}, []), ("x", "x"), ([], [])]
    got = c.ReviewBatch([upd(n, o) for n, o in cases], D.GATOR_EP)
    from parity_util import to_oracle_review
    for k in range(len(cases)):
        assert not isinstance(got[k], Exception), (k, got[k])
        want = sorted(r.msg for r in oc.review(to_oracle_review(upd(*cases[k])), D.GATOR_EP))
        assert sorted(r.msg for r in got[k]) == want
    assert [len(got[k]) for k in range(len(cases))] == [1, 0, 0, 1, 0, 1, 1]
    # the three reviews with a non-empty container in the comparison are the ones the host evaluator answered
    table = c.driver.engine.create_table([D.to_review_in(upd(n, o)) for n, o in cases], keep_docs=False)
    ev = table.eval()
    assert ev.host_evaluated == [0, 1, 2] and not ev.too_big_reviews()
    table.free()
