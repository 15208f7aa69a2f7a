"""Policy changes under load and quantifier edge cases, product vs oracle (round-2 advisor findings).

* Replacing a ConstraintTemplate while constraints of its kind exist: the constraints must follow the NEW template on
  the device plan and in the renderer (constrainttemplate_controller.go re-adds the template through
  drivers.Driver.AddTemplate, pkg/drivers/k8scel/driver.go:74-137); a replacement that does not compile leaves the old
  template serving and reports the error.
* `every x in <review ref> { .. }` over an undefined / empty / non-collection domain."""
import pytest

from gatekeeper_amd import driver as D
from gatekeeper_amd import _lib as L
from parity_util import BACKENDS, assert_parity, load_both


def _tmpl(kind, rego):
    return {"apiVersion": "templates.gatekeeper.sh/v1", "kind": "ConstraintTemplate", "metadata": {"name": kind.lower()},
            "spec": {"crd": {"spec": {"names": {"kind": kind}}}, "targets": [{"target": "admission.k8s.gatekeeper.sh", "rego": rego}]}}


def _cons(kind, name, params=None, match=None):
    spec = {}
    if params is not None:
        spec["parameters"] = params
    if match is not None:
        spec["match"] = match
    return {"apiVersion": "constraints.gatekeeper.sh/v1beta1", "kind": kind, "metadata": {"name": name}, "spec": spec}


def _cm(name, **spec):
    return {"apiVersion": "v1", "kind": "ConfigMap", "metadata": {"name": name, "namespace": "default"}, "spec": spec}


V1 = 'package k\nviolation[{"msg": "v1 a"}] { input.review.object.spec.a == true }\n'
V2 = 'package k\nviolation[{"msg": "v2 b"}] { input.review.object.spec.b == true }\n'
BROKEN = 'package k\nviolation[{"msg": "x"}] { nosuchfunction(input.review.object.spec.b) }\n'


@pytest.mark.parametrize("backend", BACKENDS)
def test_template_replaced_under_existing_constraints(backend):
    c, oc = load_both(backend, [_tmpl("KFlip", V1)], [_cons("KFlip", "one"), _cons("KFlip", "two", match={"kinds": [{"apiGroups": [""], "kinds": ["ConfigMap"]}]})])
    objs = [_cm("a", a=True), _cm("b", b=True), _cm("ab", a=True, b=True), _cm("none")]
    rv = [D.AugmentedUnstructured(D.Unstructured(o), None, "Original") for o in objs]
    assert assert_parity(c, oc, rv) == 4
    got = c.ReviewBatch(rv, D.AUDIT_EP)
    assert [sorted(r.msg for r in g) for g in got] == [["v1 a", "v1 a"], [], ["v1 a", "v1 a"], []]
    # the template is replaced: the SAME constraints now answer with v2's logic, on the device and in the renderer
    c.AddTemplate(_tmpl("KFlip", V2))
    oc.add_template(_tmpl("KFlip", V2))
    assert assert_parity(c, oc, rv) == 4
    got = c.ReviewBatch(rv, D.AUDIT_EP)
    assert [sorted(r.msg for r in g) for g in got] == [[], ["v2 b", "v2 b"], ["v2 b", "v2 b"], []]
    # native single-review path sees the new logic as well
    assert [r.msg for r in c.Review(rv[1], D.AUDIT_EP)] == ["v2 b", "v2 b"]
    # a replacement that does not compile is refused and the template in force keeps serving
    with pytest.raises(Exception):
        c.AddTemplate(_tmpl("KFlip", BROKEN))
    got = c.ReviewBatch(rv, D.AUDIT_EP)
    assert [sorted(r.msg for r in g) for g in got] == [[], ["v2 b", "v2 b"], ["v2 b", "v2 b"], []]


EVERY = ('package k\nimport future.keywords.every\nimport future.keywords.in\n'
         'violation[{"msg": "all named a"}] { every c in input.review.object.spec.containers { c.name == "a" } }\n')
EVERY_NEG = ('package k\nimport future.keywords.every\nimport future.keywords.in\n'
             'violation[{"msg": "not all a"}] { not all_a }\n'
             'all_a { every c in input.review.object.spec.containers { c.name == "a" } }\n')


@pytest.mark.parametrize("backend", BACKENDS)
def test_every_over_absent_empty_and_non_collection_domains(backend):
    c, oc = load_both(backend, [_tmpl("KEvery", EVERY), _tmpl("KEveryNeg", EVERY_NEG)], [_cons("KEvery", "e"), _cons("KEveryNeg", "n")])
    objs = [_cm("absent"), _cm("empty", containers=[]), _cm("scalar", containers="abc"), _cm("num", containers=7),
            _cm("all-a", containers=[{"name": "a"}, {"name": "a"}]), _cm("one-b", containers=[{"name": "a"}, {"name": "b"}]),
            _cm("unnamed", containers=[{"image": "x"}]), _cm("null", containers=None)]
    rv = [D.AugmentedUnstructured(D.Unstructured(o), None, "Original") for o in objs]
    assert_parity(c, oc, rv)
    got = c.ReviewBatch(rv, D.AUDIT_EP)
    by_name = {o["metadata"]["name"]: sorted(r.msg for r in g) for o, g in zip(objs, got)}
    # undefined domain: `every` is undefined (so `not all_a` holds); empty / non-collection domain: vacuously true
    assert by_name["absent"] == ["not all a"]
    assert by_name["empty"] == ["all named a"]
    assert by_name["all-a"] == ["all named a"]
    assert by_name["one-b"] == ["not all a"]
    assert by_name["unnamed"] == ["not all a"]
