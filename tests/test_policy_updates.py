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


DEEP1 = ('package k\n'
         'joined(obj) = out { out := concat(",", sort([s | s := obj.spec.names[_]; is_string(s)])) }\n'
         'violation[{"msg": msg}] { j := joined(input.review.object); j == input.parameters.want; msg := sprintf("v1 %v", [j]) }\n')
DEEP2 = ('package k\n'
         'joined(obj) = out { out := concat("+", sort([upper(s) | s := obj.spec.names[_]; is_string(s)])) }\n'
         'violation[{"msg": msg}] { j := joined(input.review.object); j == input.parameters.want; msg := sprintf("v2 %v", [j]) }\n')


@pytest.mark.parametrize("backend", BACKENDS)
def test_template_with_a_deep_helper_replaced(backend):
    """A closed helper the formula language cannot express (it sorts) is evaluated by the FLATTENER on the sub-document it reads
    (spec.names) -- a deep dictionary expression, dexpr.hpp.  Replacing the template must replace the closure: the same constraints
    answer with the new helper, on the device and in the renderer; two constraints with different parameters share the expression."""
    c, oc = load_both(backend, [_tmpl("KDeep", DEEP1)], [_cons("KDeep", "ab", {"want": "a,b"}), _cons("KDeep", "empty", {"want": ""}), _cons("KDeep", "up", {"want": "A+B"})])
    objs = [_cm("ba", names=["b", "a"]), _cm("ab7", names=["a", 7, "b"]), _cm("none"), _cm("empty", names=[]), _cm("scalar", names="a,b"), _cm("abc", names=["c", "b", "a"]),
            _cm("obj", names={"x": "b", "y": "a"})]
    rv = [D.AugmentedUnstructured(D.Unstructured(o), None, "Original") for o in objs]
    assert assert_parity(c, oc, rv) == 6
    got = c.ReviewBatch(rv, D.AUDIT_EP)
    assert [sorted(r.msg for r in g) for g in got] == [["v1 a,b"], ["v1 a,b"], ["v1 "], ["v1 "], ["v1 "], [], ["v1 a,b"]]
    c.AddTemplate(_tmpl("KDeep", DEEP2))
    oc.add_template(_tmpl("KDeep", DEEP2))
    assert assert_parity(c, oc, rv) == 6
    got = c.ReviewBatch(rv, D.AUDIT_EP)
    assert [sorted(r.msg for r in g) for g in got] == [["v2 A+B"], ["v2 A+B"], ["v2 "], ["v2 "], ["v2 "], [], ["v2 A+B"]]


FMT_MANY = 'package k\nviolation[{"msg": "m"}] { sprintf("%v%v%v%v", [input.review.name, input.review.namespace, input.review.kind.kind, input.review.kind.group]) == input.parameters.s }\n'


def test_formatted_comparison_limits_and_array_concat_of_non_arrays():
    """(1) a constant that can be cut in too many ways for adjacent verbs is refused, not approximated; (2) array.concat with a
    non-array constant operand is undefined (the rule body fails), as in OPA"""
    c, oc = load_both("hostemu", [_tmpl("KFmtMany", FMT_MANY)], [])
    with pytest.raises(D.UnsupportedError, match="too many ways to cut"):
        c.AddConstraint(_cons("KFmtMany", "x", {"s": "a" * 40}))
    c.AddConstraint(_cons("KFmtMany", "short", {"s": "abPod"}))
    oc.add_constraint(_cons("KFmtMany", "short", {"s": "abPod"}))
    pods = [{"apiVersion": "v1", "kind": "Pod", "metadata": {"name": n, "namespace": ns}} for n, ns in (("a", "b"), ("ab", ""), ("", "ab"), ("x", "y"))]
    assert assert_parity(c, oc, [D.AugmentedUnstructured(D.Unstructured(o), None, "Original") for o in pods]) >= 1
    rego = ('package k\nviolation[{"msg": "never"}] { x := array.concat(input.parameters.notarray, [o | o := input.review.object.spec.names[_]]); count(x) >= 0 }\n'
            'violation[{"msg": "found"}] { x := array.concat(["p"], [o | o := input.review.object.spec.names[_]; o != "skip"]); x[_] == input.parameters.find }\n')
    c2, oc2 = load_both("hostemu", [_tmpl("KConcat", rego)], [_cons("KConcat", "b", {"notarray": "str", "find": "b"}), _cons("KConcat", "p", {"notarray": 3, "find": "p"}),
                                                              _cons("KConcat", "skip", {"notarray": {}, "find": "skip"})])
    objs = [_cm("two", names=["a", "b"]), _cm("skip", names=["skip", "b"]), _cm("none"), _cm("one", names=["z"])]
    assert assert_parity(c2, oc2, [D.AugmentedUnstructured(D.Unstructured(o), None, "Original") for o in objs]) == 2 + 4
