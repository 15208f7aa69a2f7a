"""Builtins over review data whose DEFINEDNESS decides whether a violation exists (a Rego rule body fails when any term
is undefined): object.get with a key path, type_name, concat over an array literal.  The partial evaluator carries them
as opaque values with an exact definedness formula; the messages are rendered by the concrete evaluator.  Product vs
oracle (rendered results and raw bitmaps) on objects that hit every branch: value present / absent / of the wrong type,
an intermediate member that is not an object, a root that is not an object."""
import pytest

from gatekeeper_amd import driver as D
from parity_util import BACKENDS, assert_parity, load_both


def tmpl(kind, rego):
    return {"apiVersion": "templates.gatekeeper.sh/v1", "kind": "ConstraintTemplate", "metadata": {"name": kind.lower()},
            "spec": {"crd": {"spec": {"names": {"kind": kind}}}, "targets": [{"target": "admission.k8s.gatekeeper.sh", "rego": rego}]}}


T = {
    "K8sGetPath": '''package k
violation[{"msg": msg}] {
  policy := object.get(input.review.object, ["spec", "dnsPolicy"], "ClusterFirst")
  policy != "Default"
  msg := sprintf("dnsPolicy %v", [policy])
}
violation[{"msg": msg}] {
  v := object.get(input.review.object.metadata, ["annotations", "a/b"], "none")
  v == "none"
  msg := "annotation a/b missing"
}
violation[{"msg": msg}] {
  whole := object.get(input.review.object.spec, [], {})
  whole.hostNetwork == true
  msg := "hostNetwork via the empty path"
}
''',
    "K8sTypeName": '''package k
violation[{"msg": msg}] {
  v := input.review.object.spec.replicas
  not is_number(v)
  msg := sprintf("replicas is a %v", [type_name(v)])
}
violation[{"msg": msg}] {
  msg := sprintf("labels is a %v", [type_name(input.review.object.metadata.labels)])
}
''',
    "K8sConcat": '''package k
violation[{"msg": msg}] {
  n := input.review.object.metadata.name
  msg := concat("/", ["pods", input.review.object.metadata.namespace, lower(n)])
}
violation[{"msg": msg}] {
  msg := concat("-", ["replicas", input.review.object.spec.replicas])
}
violation[{"msg": msg}] {
  msg := concat(":", [sprintf("%v", [input.review.object.spec.replicas]), "x"])
}
''',
}


def obj(name, spec, ns="default", **md):
    m = {"name": name, "namespace": ns}
    m.update(md)
    return {"apiVersion": "v1", "kind": "Pod", "metadata": m, "spec": spec}


OBJS = [
    obj("A", {"dnsPolicy": "None", "replicas": 2, "hostNetwork": True}, labels={"x": "y"}, annotations={"a/b": "1"}),
    obj("b", {"dnsPolicy": "Default", "replicas": "two"}, labels=["not", "a", "map"]),
    obj("c", {"replicas": None}, annotations={"other": "1"}),
    obj("d", "spec-is-a-string", annotations="annotations-is-a-string"),
    obj("e", {"dnsPolicy": {"nested": True}, "replicas": 1.5, "hostNetwork": False}, ns="kube-system", labels={}),
    {"apiVersion": "v1", "kind": "Pod", "metadata": {"name": "f"}},
    {"apiVersion": "v1", "kind": "Pod", "metadata": "metadata-is-a-string", "spec": {"replicas": True}},
]


@pytest.mark.parametrize("backend", BACKENDS)
def test_definedness_of_opaque_builtin_results(backend):
    c, oc = load_both(backend, [tmpl(k, r) for k, r in T.items()],
                      [{"apiVersion": "constraints.gatekeeper.sh/v1beta1", "kind": k, "metadata": {"name": "x"}, "spec": {}} for k in T])
    reviews = [D.AugmentedUnstructured(D.Unstructured(o), None, "Original") for o in OBJS]
    n = assert_parity(c, oc, reviews, D.GATOR_EP)
    assert n > 15
    got = [sorted(r.msg for r in g) for g in c.ReviewBatch(reviews, D.GATOR_EP)]
    assert "pods/default/a" in got[0] and "dnsPolicy None" in got[0] and "hostNetwork via the empty path" in got[0]
    assert "replicas is a string" in got[1] and "replicas-two" in got[1] and "labels is a array" in got[1]
    assert not any(m.startswith("replicas-") for m in got[0])        # a number is not a string: concat is undefined
    assert "annotation a/b missing" in got[2] and "annotation a/b missing" not in got[0]
