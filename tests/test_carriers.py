"""Element carriers (round 6; csrc/plan.hpp T_ABSENT): the plans read an array element's marker (presence, parent ordinal, count) from the
rows of ONE member of the element -- `name` in every in-tree template -- instead of the element's own rows, and the flattener guarantees
one row at the carrier's path per element: the member's own, or a T_ABSENT row for an element that has no such member or is no object.
These objects are all about elements WITHOUT the carrier: the iteration `containers[_]` must still see them (Rego iterates every
element), every other predicate on the member's path must see "no row".  Product vs oracle, every backend, both ingest paths, pruned
and full tables; the raw bitmaps are compared too (assert_parity)."""
import pytest

from gatekeeper_amd import driver as D
from parity_util import BACKENDS, assert_parity, load_both, make_client


def tmpl(kind, rego):
    return {"apiVersion": "templates.gatekeeper.sh/v1", "kind": "ConstraintTemplate", "metadata": {"name": kind.lower()},
            "spec": {"crd": {"spec": {"names": {"kind": kind}}}, "targets": [{"target": D.TARGET_NAME, "rego": rego}]}}


def con(kind, name="c", params=None):
    c = {"apiVersion": "constraints.gatekeeper.sh/v1beta1", "kind": kind, "metadata": {"name": name}, "spec": {}}
    if params is not None:
        c["spec"]["parameters"] = params
    return c


TEMPLATES = [
    # the marker + a value test + a negated test + a join on the carrier member
    tmpl("CNamed", 'package k\nviolation[{"msg": msg}] {\n  c := input.review.object.spec.containers[_]\n  c.name == "bad"\n  msg := "named bad"\n}\n'),
    tmpl("CUnnamed", 'package k\nviolation[{"msg": "a container without a name"}] {\n  c := input.review.object.spec.containers[_]\n  not c.name\n}\n'),
    tmpl("CNotGood", 'package k\nviolation[{"msg": "not good"}] {\n  c := input.review.object.spec.containers[_]\n  c.name != "good"\n}\n'),
    tmpl("CPriv", 'package k\nviolation[{"msg": "privileged"}] {\n  c := input.review.object.spec.containers[_]\n  c.securityContext.privileged\n}\n'),
    tmpl("CCount", 'package k\nviolation[{"msg": msg}] {\n  n := count(input.review.object.spec.containers)\n  n > 2\n  msg := sprintf("%v containers", [n])\n}\n'),
    tmpl("CAny", 'package k\nviolation[{"msg": "has an element"}] {\n  input.review.object.spec.containers[_]\n}\n'),
    tmpl("CJoin", 'package k\nviolation[{"msg": msg}] {\n  m := input.review.object.spec.containers[_].volumeMounts[_]\n  v := input.review.object.spec.volumes[_]\n'
                  '  m.name == v.name\n  v.hostPath\n  msg := sprintf("mounts hostPath volume %v", [v.name])\n}\n'),
    tmpl("CMountRO", 'package k\nviolation[{"msg": "writable mount"}] {\n  c := input.review.object.spec.containers[_]\n  m := c.volumeMounts[_]\n  not m.readOnly\n  c.name\n}\n'),
]


def _objs():
    def pod(name, containers, volumes=None):
        spec = {"containers": containers}
        if volumes is not None:
            spec["volumes"] = volumes
        return {"apiVersion": "v1", "kind": "Pod", "metadata": {"name": name, "namespace": "d"}, "spec": spec}
    return [
        pod("all-named", [{"name": "good", "image": "i"}, {"name": "bad", "image": "i"}]),
        pod("none-named", [{"image": "i"}, {"image": "j", "securityContext": {"privileged": True}}]),
        pod("mixed", [{"image": "i"}, {"name": "good"}, {"name": "bad", "securityContext": {"privileged": True}}, {"image": "k"}]),
        pod("scalars", ["a", 3, None, True]),
        pod("nested-arrays", [[{"name": "bad"}], []]),
        pod("empty", []),
        pod("name-is-null", [{"name": None}, {"name": 7}, {"name": {"x": 1}}, {"name": []}, {"name": ""}]),
        pod("first-unnamed-priv", [{"securityContext": {"privileged": True}}, {"name": "good"}]),
        pod("join", [{"name": "c", "volumeMounts": [{"name": "v1"}, {"mountPath": "/x"}, {"name": "v2", "readOnly": True}]}, {"volumeMounts": [{"name": "v2"}]}],
            [{"name": "v1", "hostPath": {"path": "/"}}, {"hostPath": {"path": "/tmp"}}, {"name": "v2", "emptyDir": {}}]),
        pod("join-unnamed-volume", [{"name": "c", "volumeMounts": [{}, {"name": None}]}], [{"hostPath": {"path": "/"}}, {"name": None, "hostPath": {}}]),
        pod("many", [{"image": "i%d" % i} if i % 3 else {"name": "bad" if i == 9 else "good"} for i in range(40)]),
        {"apiVersion": "v1", "kind": "Pod", "metadata": {"name": "no-spec", "namespace": "d"}},
        {"apiVersion": "v1", "kind": "Pod", "metadata": {"name": "containers-is-object", "namespace": "d"}, "spec": {"containers": {}}},
    ]


@pytest.mark.parametrize("ingest", ["index", "text", "general"])
@pytest.mark.parametrize("pruned", [False, True])
@pytest.mark.parametrize("backend", BACKENDS)
def test_elements_without_the_carrier_member(backend, pruned, ingest, monkeypatch):
    if pruned:
        monkeypatch.setenv("GK_FORCE_PRUNE", "1")
    if ingest == "text":
        monkeypatch.setenv("GK_NO_INDEX", "1")
    if ingest == "general":
        monkeypatch.setenv("GK_SLOW_INGEST", "1")
    c, oc = load_both(backend, TEMPLATES, [con(t["spec"]["crd"]["spec"]["names"]["kind"]) for t in TEMPLATES])
    reviews = [D.AugmentedUnstructured(D.Unstructured(o), None, "Original") for o in _objs()]
    assert assert_parity(c, oc, reviews, D.GATOR_EP) > 25


def test_the_plans_read_member_rows_not_element_rows():
    """what the feature is for: with `name` as the carrier no plan path is the bare element `containers[]` any more"""
    c = make_client("hostemu")
    for t in TEMPLATES[:4]:
        c.AddTemplate(t)
        c.AddConstraint(con(t["spec"]["crd"]["spec"]["names"]["kind"]))
    table = c.driver.engine.create_table([D.to_review_in(D.AugmentedUnstructured(D.Unstructured(o), None, "Original")) for o in _objs()])
    try:
        ev = table.eval()
        # rows read: one per container (carrier rows, real or T_ABSENT) + the privileged rows -- not two per container
        n_containers = sum(len(o.get("spec", {}).get("containers") or []) if isinstance(o.get("spec", {}).get("containers"), list) else 0 for o in _objs())
        assert n_containers <= ev.n_rows_read < 2 * n_containers
    finally:
        table.free()


# ---- a T_ABSENT row sits on its path for EVERY plan of the engine, not only for the one that made the path a carrier (found by
# tests/test_template_fuzz.py on the generated code, round 6)
T_GT = tmpl("IGt", 'package k\nviolation[{"msg": "a > 2"}] {\n  e := input.review.object.spec.items[_]\n  e.a > 2\n}\n')                     # makes `a` the carrier of spec.items[]
T_ANY = tmpl("IAnyA", 'package k\nviolation[{"msg": "some item has a"}] {\n  input.review.object.spec.items[_].a\n}\n')                       # a flat wildcard test on the carrier's path
T_B = tmpl("IB", 'package k\nviolation[{"msg": msg}] {\n  e := input.review.object.spec.items[_]\n  e.b == 1\n  msg := "b is 1"\n}\n')       # iterates the same elements, never looks at `a`
T_NOT = tmpl("INotA", 'package k\nviolation[{"msg": "an item without a"}] {\n  e := input.review.object.spec.items[_]\n  not e.a\n}\n')


def _items(name, items):
    return {"apiVersion": "v1", "kind": "Pod", "metadata": {"name": name, "namespace": "d"}, "spec": {"items": items}}


ITEM_OBJS = [_items("none", [{"b": 1}]), _items("one", [{"a": 3, "b": 1}]), _items("mixed", [{"b": 2}, {"a": 1}, {"b": 1}, 7, []]), _items("empty", []),
             _items("falsy", [{"a": False}, {"a": None}]), _items("nested", [{"a": {"a": 5}}, {"b": {"a": 3}}])]


@pytest.mark.parametrize("group_max", [0, 1], ids=["one-plan", "a-plan-per-constraint"])
@pytest.mark.parametrize("backend", BACKENDS)
def test_absent_rows_are_no_rows_to_other_predicates_and_other_plans(backend, group_max):
    templates = [T_GT, T_ANY, T_B, T_NOT]
    from parity_util import plan_group_max
    from gatekeeper_amd import _lib
    with plan_group_max(_lib.load(hostemu=backend.startswith("hostemu")), group_max):
        c, oc = load_both(backend, templates, [con(t["spec"]["crd"]["spec"]["names"]["kind"]) for t in templates])
        reviews = [D.AugmentedUnstructured(D.Unstructured(o), None, "Original") for o in ITEM_OBJS]
        assert assert_parity(c, oc, reviews, D.GATOR_EP) >= 8
