"""The host renderer's two evaluators (csrc/ceval.cpp: concrete, tried first; csrc/pe.cpp: the partial evaluator on a concrete
review, the fallback): the same violation sets -- tests/conftest.py switches the cross-check on for the whole suite -- and the
concrete one really is the one that serves (a silent fallback for everything would make the cross-check vacuous)."""
import os

import pytest
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

SCRIPT = r"""
import sys
from gatekeeper_amd import driver as D, synth
fx = synth.load_fixtures()
drv = D.Driver(device=0, hostemu=True); client = D.Client(drv)
t, c = synth.corpus(fx)
for x in t[::3]: client.AddTemplate(x)
kinds = {x["spec"]["crd"]["spec"]["names"]["kind"] for x in t[::3]}
for x in c:
    if x["kind"] in kinds: client.AddConstraint(x)
for x in synth.psp_templates(fx): client.AddTemplate(x)
for x in synth.audit_constraints(): client.AddConstraint(x)
n = 300
nss = synth.gen_namespaces()
b = synth.NativeBatch(drv.engine.lib, n, seed=synth.SEED, mixed=True, start=0, namespaces=nss)
tb = drv.engine.create_table_native(b.reviews, n, keep_docs=True, resident=True)
ev = tb.eval()
pairs = 0
for row, cid in enumerate(ev.constraint_ids):
    for r in D.EvalResult.bits(ev.viol[row], ev.n_reviews):
        assert tb.render(int(cid), int(r))
        pairs += 1
print("PAIRS", pairs)
"""


def test_the_concrete_evaluator_serves_and_agrees_with_the_partial_one():
    env = dict(os.environ, GK_RENDER_CHECK="1", GK_RENDER_STATS="1", PYTHONPATH=ROOT)
    p = subprocess.run([sys.executable, "-c", SCRIPT], env=env, capture_output=True, text=True, timeout=600)
    assert p.returncode == 0, p.stderr[-2000:]
    pairs = int(p.stdout.split("PAIRS")[1].split()[0])
    assert pairs > 1500
    stats = [l for l in p.stderr.splitlines() if "[gkgpu render]" in l]
    assert stats, "no statistics line: nothing was rendered through Template::render"
    last = stats[-1].split()
    fast, slow = int(last[last.index("evaluator") + 1]), int(last[len(last) - 1 - last[::-1].index("evaluator") + 1])
    assert fast >= 1024 and slow * 50 <= fast, stats[-1]


# ---- Template::render directly (tests/native/pe_cli.cpp linked against the CPU build of the engine): constructs the DEVICE plan
# refuses can still be rendered, so the evaluators meet them here -- against the oracle's interpreter, with the cross-check on, and
# with the statistics saying that the concrete evaluator served (a fallback would pass the comparison silently)
CASES = [
    ("default_else", """
package t
default limit = 3
limit = n { n := input.parameters.limit }
grade(x) = "high" { x > 10 } else = "mid" { x > 5 } else = "low" { true }
violation[{"msg": msg, "details": {"g": g}}] {
  c := input.review.object.spec.containers[_]
  count(c.ports) > limit
  g := grade(count(c.ports))
  msg := sprintf("%v has %v ports (%v), limit %v", [c.name, count(c.ports), g, limit])
}""", [{}, {"limit": 1}]),
    ("every_somein_patterns", """
package t
violation[{"msg": msg}] {
  some i, c in input.review.object.spec.containers
  every p in c.ports { p.containerPort > 1000 }
  [first, second] := [c.name, i]
  {"name": nm, "image": img} := {"name": c.name, "image": c.image}
  msg := sprintf("%v/%v/%v/%v", [first, second, nm, img])
}""", [{}]),
    ("comprehensions_sets_objects", """
package t
names := {c.name | c := input.review.object.spec.containers[_]}
by_name[n] = img { c := input.review.object.spec.containers[_]; n := c.name; img := c.image }
ports := [p.containerPort | p := input.review.object.spec.containers[_].ports[_]]
tags := {n: count(split(img, ":")) | img := by_name[n]}
violation[{"msg": msg, "details": {"tags": tags, "ports": ports}}] {
  wanted := {x | x := input.parameters.names[_]}
  missing := wanted - names
  extra := names & {"c0", "zz"}
  count(missing) + count(extra) > 0
  "c0" in names
  msg := sprintf("missing %v extra %v all %v", [missing, extra, names | wanted])
}""", [{"names": ["c0", "nope"]}, {"names": []}]),
    ("functions_reordering_negation", """
package t
unit("k") = 1000 { true }
unit("M") = 1000000 { true }
unit("") = 1 { true }
scaled(s) = n { suffix := substring(s, count(s) - 1, -1); m := unit(suffix); n := to_number(trim_suffix(s, suffix)) * m }
scaled(s) = n { not unit(substring(s, count(s) - 1, -1)); n := to_number(s) }
is_exempt(c) { input.parameters.exempt[_] == c.name }
violation[{"msg": msg}] {
  msg := sprintf("%v: %v > %v", [c.name, got, cap])
  got > cap
  cap := scaled(input.parameters.cap)
  got := scaled(c.size)
  not is_exempt(c)
  c := input.review.object.spec.containers[_]
}""", [{"cap": "2k", "exempt": ["c1"]}, {"cap": "1", "exempt": []}]),
    ("keys_wildcards_nested", """
package t
violation[{"msg": msg}] {
  v := input.review.object.metadata.labels[k]
  allowed := input.parameters.allowed[k][_]
  not v == allowed
  count({a | a := input.parameters.allowed[k][_]; a == v}) == 0
  msg := sprintf("label %v=%v not in %v", [k, v, input.parameters.allowed[k]])
}
violation[{"msg": "no labels"}] { not input.review.object.metadata.labels }
violation[{"msg": msg}] {
  input.review.object.spec.containers[_].ports[_].containerPort == input.parameters.banned[_]
  msg := "banned port"
}""", [{"allowed": {"app": ["web", "db"], "tier": ["x"]}, "banned": [80, 8080]}, {"allowed": {}, "banned": []}]),
]

OBJECTS = [
    {"apiVersion": "v1", "kind": "Pod", "metadata": {"name": "p0", "labels": {"app": "cache", "tier": "x"}},
     "spec": {"containers": [{"name": "c0", "image": "nginx:1.2", "size": "3k", "ports": [{"containerPort": 80}, {"containerPort": 8443}]},
                             {"name": "c1", "image": "busybox", "size": "5M", "ports": [{"containerPort": 2000}, {"containerPort": 3000}, {"containerPort": 4000}, {"containerPort": 5000},
                                                                                          {"containerPort": 6000}, {"containerPort": 7000}]}]}},
    {"apiVersion": "v1", "kind": "Pod", "metadata": {"name": "p1"}, "spec": {"containers": [{"name": "zz", "image": "a:b:c", "size": "7", "ports": []}]}},
    {"apiVersion": "v1", "kind": "Pod", "metadata": {"name": "p2", "labels": {}}, "spec": {"containers": []}},
]


@pytest.fixture(scope="module")
def pe_cli(tmp_path_factory):
    exe = str(tmp_path_factory.mktemp("pecli") / "pe_cli")
    native = os.path.join(ROOT, "tests", "native")
    subprocess.run(["g++", "-O1", "-std=c++17", "-I", os.path.join(ROOT, "gatekeeper_amd", "csrc"), os.path.join(native, "pe_cli.cpp"), "-o", exe,
                    "-L", native, "-lgkgpu_hostemu", "-Wl,-rpath," + native, "-lpthread"], check=True, timeout=300)
    return exe


@pytest.mark.parametrize("name,rego,params", CASES, ids=[c[0] for c in CASES])
def test_render_of_constructs_the_device_plan_may_refuse(name, rego, params, pe_cli):
    import json
    from oracle.rego_interp import Interp
    from oracle.values import from_json, to_json
    ip = Interp([rego])
    served_fast = served_slow = n_results = 0
    for p in params:
        for obj in OBJECTS:
            review = {"object": obj, "kind": {"group": "", "version": "v1", "kind": "Pod"}, "name": obj["metadata"]["name"], "operation": "CREATE"}
            want = sorted(json.dumps(to_json(v), sort_keys=True) for v in ip.violations(from_json({"review": review, "parameters": p})))
            env = dict(os.environ, GK_RENDER_CHECK="1", GK_RENDER_STATS="1")
            r = subprocess.run([pe_cli], input=json.dumps({"rego": rego, "parameters": p, "review": review}), env=env, capture_output=True, text=True, timeout=60)
            out = json.loads(r.stdout)
            assert "error" not in out, (name, out, r.stderr[-500:])
            got = sorted(json.dumps(dict({"msg": v["msg"]}, **({"details": v["details"]} if "details" in v else {})), sort_keys=True) for v in out["violations"])
            want_msgs = sorted(json.dumps({k: v for k, v in json.loads(w).items() if k in ("msg", "details")}, sort_keys=True) for w in want)
            assert got == sorted(set(want_msgs)), (name, p, obj["metadata"]["name"])
            n_results += len(got)
            served_slow += sum(1 for l in r.stderr.splitlines() if "[gkgpu render]" in l and "partial evaluator 0 " not in l)
            served_fast += 1
    assert n_results >= 2, "the case renders nothing: it tests nothing"
    assert served_slow == 0, "the concrete evaluator fell back on %d of %d calls of %s" % (served_slow, served_fast, name)
