"""RESULT totals (pkg/audit/manager.go:893-904: totalViolationsPerConstraint counts types.Results, not violating pairs) with
the device deciding which violating pairs CAN have more than one result (Template::compile_multi -> the totals plans,
gk_table_totals): a pair it does not flag counts one result unrendered, the flagged ones are rendered on the host.  Product vs
oracle on the shapes where the two numbers differ -- several rule bodies, several elements of an iterated array, nested
iterations, unrolled parameter alternatives whose conditions are identical, equal messages that collapse in the set -- and on
the shape the counting loops cannot walk (an OBJECT where the template iterates elements: the totals plan refuses the review
locally and it is rendered, while the violation bitmap is still answered on the device)."""
import os

import pytest

from gatekeeper_amd import driver as D
from oracle import client as OC
from oracle import target as OT
from parity_util import BACKENDS, make_client


def tmpl(kind, rego):
    return {"apiVersion": "templates.gatekeeper.sh/v1", "kind": "ConstraintTemplate", "metadata": {"name": kind.lower()},
            "spec": {"crd": {"spec": {"names": {"kind": kind}}}, "targets": [{"target": "admission.k8s.gatekeeper.sh", "rego": rego}]}}


T = {
    # one body, one result per violating CONTAINER (the PSP shape)
    "K8sPerElement": ('''package k
violation[{"msg": msg}] {
  c := input.review.object.spec.containers[_]
  c.securityContext.privileged
  msg := sprintf("privileged container %v", [c.name])
}
''', {}),
    # two bodies over the same review: none, one or both hold
    "K8sTwoBodies": ('''package k
violation[{"msg": msg}] {
  input.review.object.spec.hostNetwork
  msg := "hostNetwork"
}
violation[{"msg": msg}] {
  input.review.object.spec.hostPID
  msg := "hostPID"
}
''', {}),
    # nested iteration: a result per (container, port)
    "K8sNested": ('''package k
violation[{"msg": msg}] {
  c := input.review.object.spec.containers[_]
  p := c.ports[_]
  p.hostPort > 0
  msg := sprintf("%v uses host port %v", [c.name, p.hostPort])
}
''', {}),
    # the message does not depend on the element: equal messages of different elements are ONE member of the set
    "K8sSameMessage": ('''package k
violation[{"msg": msg}] {
  c := input.review.object.spec.containers[_]
  c.securityContext.privileged
  msg := "some container is privileged"
}
''', {}),
    # parameter alternatives with IDENTICAL conditions and different messages: one result per alternative
    "K8sParamAlternatives": ('''package k
violation[{"msg": msg}] {
  l := input.parameters.labels[_]
  not input.review.object.metadata.labels.team
  msg := sprintf("label team missing (reported for %v)", [l])
}
''', {"labels": ["a", "b", "c"]}),
    # an element joined with a value outside its array + details in the head
    "K8sDetails": ('''package k
violation[{"msg": msg, "details": {"c": c.name}}] {
  c := input.review.object.spec.containers[_]
  not startswith(c.image, "good/")
  msg := "image from a repository that is not allowed"
}
''', {}),
}


def pod(name, containers, **spec):
    s = dict(spec)
    s["containers"] = containers
    return {"apiVersion": "v1", "kind": "Pod", "metadata": {"name": name, "namespace": "default", "labels": spec.pop("labels", {"app": name})}, "spec": s}


def ctr(name, priv=False, image="good/x", ports=()):
    c = {"name": name, "image": image}
    if priv:
        c["securityContext"] = {"privileged": True}
    if ports:
        c["ports"] = [{"hostPort": p, "containerPort": 80} for p in ports]
    return c


OBJS = [
    pod("clean", [ctr("a")]),
    pod("one", [ctr("a", priv=True), ctr("b")], hostNetwork=True),
    pod("two", [ctr("a", priv=True, image="bad/x"), ctr("b", priv=True, image="bad/y", ports=(8080, 9090))], hostNetwork=True, hostPID=True),
    pod("same-names", [ctr("a", priv=True, image="bad/x"), ctr("a", priv=True, image="bad/x", ports=(1,))]),      # equal messages collapse
    pod("three", [ctr("a", ports=(1, 2)), ctr("b", ports=(3,)), ctr("c", priv=True, image="worse/z")], hostPID=True),
    # an OBJECT where the templates iterate elements: Rego walks its values
    {"apiVersion": "v1", "kind": "Pod", "metadata": {"name": "object-containers", "namespace": "default"},
     "spec": {"containers": {"x": ctr("a", priv=True, image="bad/1"), "y": ctr("b", priv=True, image="bad/2")}}},
]


FLAT = {
    # reaches the containers through ONE wildcard predicate: the violation formula needs no element loop, hence no guard on
    # spec.containers -- counting its bindings does
    "K8sFlat": ('''package k
violation[{"msg": msg}] {
  input.review.object.spec.containers[_].securityContext.privileged
  msg := "privileged"
}
''', {}),
    "K8sTwoBodies": T["K8sTwoBodies"],
}


def run_totals(backend, templates, objs):
    c = make_client(backend)
    oc = OC.Client()
    for kind, (rego, params) in sorted(templates.items()):
        k = {"apiVersion": "constraints.gatekeeper.sh/v1beta1", "kind": kind, "metadata": {"name": kind.lower()}, "spec": {"parameters": params}}
        c.AddTemplate(tmpl(kind, rego)); c.AddConstraint(k)
        oc.add_template(tmpl(kind, rego)); oc.add_constraint(k)
    rins = [D.to_review_in(D.AugmentedUnstructured(D.Unstructured(o), None, "Original"), None) for o in objs]
    table = c.driver.engine.create_table(rins, resident=True, process="audit")
    try:
        ev = table.eval()
        refused = set(int(r) for r in ev.too_big_reviews())
        got = table.totals()
        rendered = table.rendered_pairs
        os.environ["GK_TOTALS_RENDER_ALL"] = "1"
        try:
            every = table.totals()
            rendered_all = table.rendered_pairs
        finally:
            del os.environ["GK_TOTALS_RENDER_ALL"]
    finally:
        table.free()
    want, want_pairs = {}, {}
    for i, o in enumerate(objs):
        if i in refused:
            continue   # beyond the engine's limits: reported, the caller fails closed (no bits, no results)
        for r in oc.review(OT.AugmentedUnstructured(OT.Unstructured(o), None, "Original"), OC.AUDIT_EP):
            key = r.constraint["kind"]
            want[key] = want.get(key, 0) + 1
            want_pairs.setdefault(key, set()).add(o["metadata"]["name"])
    kind_of = {drv_id: rec[0].get("kind") for drv_id, rec in c._active(D.AUDIT_EP).items()}
    results = {kind_of[cid]: v[0] for cid, v in got.items() if v[0]}
    pairs = {kind_of[cid]: v[1] for cid, v in got.items() if v[1]}
    assert results == want
    assert pairs == {k: len(v) for k, v in want_pairs.items()}
    assert got == every                                   # ... and the host pass over every violating pair says the same
    assert rendered_all == sum(pairs.values())
    return refused, want, want_pairs, rendered, rendered_all


@pytest.mark.parametrize("backend", BACKENDS)
def test_result_totals_counted_on_the_device_equal_the_oracle(backend):
    refused, want, want_pairs, rendered, rendered_all = run_totals(backend, T, OBJS)
    # (the templates that join an element's fields loop over spec.containers in their VIOLATION formulas: the object in its place
    #  is beyond the DEVICE's limits for them -- answered by the host evaluator since round 5, its results part of the totals)
    assert refused == set()
    # the workload separates results from pairs in every template but the one whose messages collapse
    assert want["K8sPerElement"] > len(want_pairs["K8sPerElement"]) and want["K8sSameMessage"] == len(want_pairs["K8sSameMessage"])
    assert want["K8sParamAlternatives"] == 3 * len(want_pairs["K8sParamAlternatives"])
    assert rendered < rendered_all                        # the device answered part of them


@pytest.mark.parametrize("backend", BACKENDS)
def test_an_object_where_bindings_are_counted_is_refused_by_the_totals_plan_only(backend):
    refused, want, want_pairs, rendered, rendered_all = run_totals(backend, FLAT, OBJS)
    assert not refused                      # the violation bitmap of every review comes from the device ...
    assert "object-containers" in want_pairs["K8sFlat"]
    assert 0 < rendered < rendered_all      # ... the object's result count from the renderer, the single-binding pairs from the device


# ---- round 4: the result COUNT on the device (Template::count_forms).  A branch with one iteration over array elements whose
# message starts "<text><%v of a leaf of the element><text>.." counts one result per firing element -- thresholds "at least k
# elements fire" -- provided the leaves are strings, free of the first character of the text behind them and pairwise different
# (review.$dup); everything that argument does not cover is flagged and rendered.  Each case below sits on one edge of it.
COUNTED = {
    # the PSP / library shape: "<name>" leads the message.  Two rule bodies print the SAME message (merged: one result per element)
    "K8sKeyed": ('''package k
violation[{"msg": msg}] {
  c := input.review.object.spec.containers[_]
  c.securityContext.privileged
  msg := sprintf("container <%v> is privileged (image %v)", [c.name, c.image])
}
violation[{"msg": msg}] {
  c := input.review.object.spec.containers[_]
  c.securityContext.allowPrivilegeEscalation
  msg := sprintf("container <%v> is privileged (image %v)", [c.name, c.image])
}
''', {}),
    # unrolled parameter alternatives that differ in ONE constant operand + a second format whose text behind the key differs
    "K8sProbes": ('''package k
violation[{"msg": msg}] {
  c := input.review.object.spec.containers[_]
  p := input.parameters.probes[_]
  not c[p]
  msg := sprintf("Container <%v> in your <%v> has no <%v>", [c.name, input.review.kind.kind, p])
}
violation[{"msg": msg}] {
  c := input.review.object.spec.containers[_]
  not c.resources
  msg := sprintf("Container <%v> has no resources", [c.name])
}
''', {"probes": ["readinessProbe", "livenessProbe"]}),
    # the array is named by an iterated, pinned key (K8sContainerLimits' spec[field][_]): one branch per pinned member
    "K8sFields": ('''package k
violation[{"msg": msg}] {
  field := {"containers", "initContainers"}[_]
  c := input.review.object.spec[field][_]
  not startswith(c.image, "good/")
  msg := sprintf("container <%v> has a bad image", [c.name])
}
''', {}),
}


def cpod(name, containers, init=()):
    s = {"containers": containers}
    if init:
        s["initContainers"] = list(init)
    return {"apiVersion": "v1", "kind": "Pod", "metadata": {"name": name, "namespace": "default"}, "spec": s}


def cc(name, priv=False, esc=False, image="good/x", probes=(), resources=True):
    c = {"name": name, "image": image}
    sc = {}
    if priv:
        sc["privileged"] = True
    if esc:
        sc["allowPrivilegeEscalation"] = True
    if sc:
        c["securityContext"] = sc
    for p in probes:
        c[p] = {"httpGet": {"path": "/"}}
    if resources:
        c["resources"] = {"limits": {"cpu": "1"}}
    return c


COUNT_OBJS = [
    cpod("clean", [cc("a", probes=("readinessProbe", "livenessProbe"))]),
    cpod("three", [cc("a", priv=True, image="bad/1"), cc("b", priv=True, esc=True, image="bad/2", resources=False), cc("c", esc=True, probes=("livenessProbe",))]),
    cpod("both-arrays", [cc("a", image="bad/1"), cc("b")], init=[cc("i0", image="bad/2"), cc("i1", image="bad/3")]),
    cpod("five", [cc("c%d" % i, priv=True, image="bad/%d" % i, resources=False) for i in range(5)]),
    cpod("seven", [cc("c%d" % i, priv=True) for i in range(7)]),                                    # more firing elements than thresholds
    cpod("dup-names", [cc("a", priv=True, image="bad/1"), cc("a", priv=True, image="bad/2")]),                     # equal keys: review.$dup
    cpod("dup-across", [cc("a", image="bad/1")], init=[cc("a", image="bad/1")]),                                     # ... across the two arrays
    cpod("separator", [cc("x> is privileged (image y", priv=True, image="bad/1"), cc("b", priv=True)]),            # the key holds the text's first character
    cpod("number-name", [cc(7, priv=True), cc("7", priv=True)]),                                                     # 7 and "7" print alike
    cpod("no-name", [{"image": "bad/1", "securityContext": {"privileged": True}}, cc("b", priv=True)]),              # undefined key: no message at all
] + [cpod("plain-%d" % i, [cc("m%d-%d" % (i, j), priv=j % 2 == 0, esc=j % 3 == 0, image=("bad/%d" % j) if (i + j) % 2 else "good/x", resources=j != 1) for j in range(2 + i % 3)],
          init=[cc("init%d" % i, image="bad/i")] if i % 2 else ()) for i in range(6)]


@pytest.mark.parametrize("backend", BACKENDS)
def test_result_counts_on_the_device_at_the_edges_of_the_argument(backend):
    refused, want, want_pairs, rendered, rendered_all = run_totals(backend, COUNTED, COUNT_OBJS)
    assert not refused
    for kind in COUNTED:
        assert want[kind] > len(want_pairs[kind])          # several results per pair everywhere
    # rendered: only what the argument does not cover -- too many elements, equal keys, the separator inside a key, a key that
    # is no string; the rest (most pairs, all of them with several results) was COUNTED
    # (seven and five -- the default is four thresholds, GK_COUNT_KMAX --: 2 pairs each; dup-names: 3; dup-across: 2; separator: 3;
    #  number-name: 2)
    assert 0 < rendered <= 15 and rendered_all >= rendered + 10   # (five: all three constraints)


def test_counting_is_switched_off_by_its_knob(monkeypatch):
    """GK_COUNT_KMAX=0: round 3's decision ("can this pair have more than one result?") serves alone -- same totals, more rendering"""
    _, _, _, rendered_counted, _ = run_totals("hostemu", COUNTED, COUNT_OBJS)
    monkeypatch.setenv("GK_COUNT_KMAX", "0")
    _, _, _, rendered_multi, _ = run_totals("hostemu", COUNTED, COUNT_OBJS)
    assert rendered_counted < rendered_multi


# members of the violation set that differ in `details` ONLY are different results (the set holds {msg, details} objects; the
# driver reports one types.Result per member, pkg/audit/manager.go:902 counts each): round-4 advisor finding -- branches were
# merged by the message's signature alone and three unrolled parameter alternatives with one constant message counted once
DETAILS = {
    # one constant message, details from an unrolled parameter: a result per label
    "K8sDetailsPerParam": ('''package k
violation[{"msg": "label team missing", "details": {"label": l}}] {
  l := input.parameters.labels[_]
  not input.review.object.metadata.labels.team
}
''', {"labels": ["a", "b", "c"]}),
    # two bodies that print the same formatted message with different constant details
    "K8sDetailsTwoBodies": ('''package k
violation[{"msg": msg, "details": {"why": "network"}}] {
  input.review.object.spec.hostNetwork
  msg := sprintf("pod %v shares a host namespace", [input.review.object.metadata.name])
}
violation[{"msg": msg, "details": {"why": "pid"}}] {
  input.review.object.spec.hostPID
  msg := sprintf("pod %v shares a host namespace", [input.review.object.metadata.name])
}
''', {}),
    # ... the SAME constant details: one member however many bodies produce it
    "K8sDetailsEqual": ('''package k
violation[{"msg": "shares a host namespace", "details": {"n": 1}}] { input.review.object.spec.hostNetwork }
violation[{"msg": "shares a host namespace", "details": {"n": 1}}] { input.review.object.spec.hostPID }
''', {}),
    # details that depend on the review, one constant message, two bodies: members differ iff the details do
    "K8sDetailsFromReview": ('''package k
violation[{"msg": "host namespace", "details": {"v": input.review.object.spec.hostNetwork}}] { input.review.object.spec.hostNetwork }
violation[{"msg": "host namespace", "details": {"v": input.review.object.spec.hostPID}}] { input.review.object.spec.hostPID }
''', {}),
    # absent details and an explicit {} are the same member (the driver reports {} for both)
    "K8sDetailsAbsentOrEmpty": ('''package k
violation[{"msg": "host namespace"}] { input.review.object.spec.hostNetwork }
violation[{"msg": "host namespace", "details": {}}] { input.review.object.spec.hostPID }
''', {}),
}


@pytest.mark.parametrize("backend", BACKENDS)
def test_members_that_differ_in_details_only_are_counted_apart(backend):
    objs = [
        pod("clean", [ctr("a")], labels={"team": "x"}),
        pod("no-team", [ctr("a")]),
        pod("net", [ctr("a")], hostNetwork=True),
        pod("pid", [ctr("a")], hostPID=True, labels={"team": "x"}),
        pod("both", [ctr("a")], hostNetwork=True, hostPID=True),
        pod("both-with-team", [ctr("a")], hostNetwork=True, hostPID=True, labels={"team": "x"}),
    ]
    refused, want, want_pairs, rendered, rendered_all = run_totals(backend, DETAILS, objs)
    assert not refused
    assert want["K8sDetailsPerParam"] == 3 * len(want_pairs["K8sDetailsPerParam"])       # three labels, three results
    assert want["K8sDetailsTwoBodies"] == len(want_pairs["K8sDetailsTwoBodies"]) + 2     # "both" and "both-with-team" yield two
    assert want["K8sDetailsEqual"] == len(want_pairs["K8sDetailsEqual"])                 # equal members collapse
    assert want["K8sDetailsFromReview"] == len(want_pairs["K8sDetailsFromReview"])       # {"v": true} twice is one member
