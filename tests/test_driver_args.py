"""The rego.Arg surface the reference constructs its driver with, at the C boundary (gk_opts) and in the mirror (driver.Driver):

  rego.Externs("inventory") | rego.Externs()      --enable-referential-rules (main.go:127,479-484): without the extern a template
                                                   that reads data.inventory is refused when it is ADDED, never compiled
  rego.DisableBuiltins(names...)                   main.go:424 (--disable-opa-builtin, default http.send; test/bats/test.bats:492-498 pins
                                                   "undefined function http.send")
  rego.GatherStats()                               pkg/gator/test/test.go:175-182; shape pinned by pkg/gator/test/test_test.go:332-418
  rego.Tracing(true)                               pkg/gator/test/test.go:184-190; test_test.go:~325 requires a non-nil Trace
The framework's own wording of the referential error is third-party (frameworks, absent): only the refusal is pinned here."""
import pytest

from gatekeeper_amd import driver as D
from gatekeeper_amd import synth

REFERENTIAL = '''package k8suniquename
violation[{"msg": msg}] {
  other := data.inventory.namespace[ns][_]["Service"][name]
  name == input.review.object.metadata.name
  ns != input.review.object.metadata.namespace
  msg := sprintf("service name %v is taken in %v", [name, ns])
}
'''
HTTP = '''package k8shttp
violation[{"msg": "x"}] { http.send({"method": "get", "url": "http://example"}).status_code == 200 }
'''
GLOB = '''package k8sglob
violation[{"msg": "x"}] { glob.match("a*", [], input.review.object.metadata.name) }
'''
NEVER = '''package nevervalidate
violation[{"msg": "never validate"}] { true }
'''


def tmpl(kind, rego):
    return {"apiVersion": "templates.gatekeeper.sh/v1", "kind": "ConstraintTemplate", "metadata": {"name": kind.lower()},
            "spec": {"crd": {"spec": {"names": {"kind": kind}}}, "targets": [{"target": "admission.k8s.gatekeeper.sh", "rego": rego}]}}


def test_referential_templates_are_refused_without_the_inventory_extern():
    on = D.Driver(device=0, hostemu=True)                          # the deployment's default: --enable-referential-rules=true
    on.AddTemplate(tmpl("K8sUniqueName", REFERENTIAL))
    off = D.Driver(device=0, hostemu=True, referential=False)      # rego.Externs(): no "inventory"
    with pytest.raises(D.EngineError) as ei:
        off.AddTemplate(tmpl("K8sUniqueName", REFERENTIAL))
    assert ei.value.code == D.L.GK_ERR_REGO and "data.inventory" in str(ei.value)
    off.AddTemplate(tmpl("NeverValidate", NEVER))                   # ... everything else is served as before


def test_disabled_builtins_are_undefined_functions():
    d = D.Driver(device=0, hostemu=True)                            # default: {"http.send"} is disabled (test/bats/test.bats:492-498)
    with pytest.raises(D.EngineError, match="undefined function http.send") as ei:
        d.AddTemplate(tmpl("K8sHttp", HTTP))
    assert ei.value.code == D.L.GK_ERR_REGO
    # with an explicit list http.send is a capability again -- valid Rego this engine does not implement: unsupported, not a type error
    d2 = D.Driver(device=0, hostemu=True, disabled_builtins=["glob.match"])
    with pytest.raises(D.UnsupportedError):
        d2.AddTemplate(tmpl("K8sHttp", HTTP))
    with pytest.raises(D.EngineError, match="undefined function glob.match") as ei:
        d2.AddTemplate(tmpl("K8sGlob", GLOB))
    assert ei.value.code == D.L.GK_ERR_REGO
    with pytest.raises(D.UnsupportedError):
        d.AddTemplate(tmpl("K8sGlob", GLOB))                         # (not disabled there: a builtin OPA has and this engine lacks)
    d3 = D.Driver(device=0, hostemu=True, disabled_builtins=[])       # nothing disabled at all
    with pytest.raises(D.UnsupportedError):
        d3.AddTemplate(tmpl("K8sHttp", HTTP))


def _never_validate(**kw):
    drv = D.Driver(device=0, hostemu=True, **kw)
    c = D.Client(drv)
    c.AddTemplate(tmpl("NeverValidate", NEVER))
    c.AddConstraint({"apiVersion": "constraints.gatekeeper.sh/v1beta1", "kind": "NeverValidate", "metadata": {"name": "always-fail"}, "spec": {}})
    obj = {"apiVersion": "v1", "kind": "Object", "metadata": {"name": "object"}}
    return drv, c, D.AugmentedUnstructured(D.Unstructured(obj), None, "Original")


def test_gather_stats_has_the_rego_drivers_shape():
    """Test_Test_withStats (pkg/gator/test/test_test.go:332-418): one entry per review and template -- scope "template", statsFor the kind,
    templateRunTimeNS (non-zero) and constraintCount (1) from source {engine, Rego}, labels TracingEnabled / PrintEnabled / target"""
    drv, c, rv = _never_validate(gather_stats=True)
    resp = drv.QueryMatching(D.TARGET_NAME, list(c.constraints.values()), rv)
    assert [r.msg for r in resp.results] == ["never validate"]
    src = {"type": "engine", "value": "Rego"}
    assert len(resp.stats_entries) == 1
    e = resp.stats_entries[0]
    assert e["scope"] == "template" and e["statsFor"] == "NeverValidate" and len(e["stats"]) == 2
    assert e["stats"][0]["name"] == "templateRunTimeNS" and e["stats"][0]["value"] != 0 and e["stats"][0]["source"] == src
    assert e["stats"][1] == {"name": "constraintCount", "value": 1, "source": src}
    assert e["labels"] == [{"name": "TracingEnabled", "value": False}, {"name": "PrintEnabled", "value": False}, {"name": "target", "value": "admission.k8s.gatekeeper.sh"}]
    assert resp.trace is None
    for s in e["stats"]:
        assert drv.GetDescriptionForStat(s["name"])
    plain, c2, rv2 = _never_validate()
    assert not plain.QueryMatching(D.TARGET_NAME, list(c2.constraints.values()), rv2).stats_entries       # no GatherStats, no stats


def test_tracing_returns_a_trace():
    drv, c, rv = _never_validate(tracing=True)                       # rego.Tracing(true): every Query carries one
    resp = drv.QueryMatching(D.TARGET_NAME, list(c.constraints.values()), rv)
    assert resp.trace and "NeverValidate/always-fail" in resp.trace and "never validate" in resp.trace and "on the device" in resp.trace
    plain, c2, rv2 = _never_validate()
    assert plain.QueryMatching(D.TARGET_NAME, list(c2.constraints.values()), rv2).trace is None
    asked = plain.QueryMatching(D.TARGET_NAME, list(c2.constraints.values()), rv2, tracing=True)            # ... or the review asks for it
    assert asked.trace and "1 result(s)" in asked.trace
    # a review the host evaluator answers says so
    huge = {"apiVersion": "v1", "kind": "Pod", "metadata": {"name": "huge"}, "spec": {"containers": [{"name": "c%d" % i, "image": "x"} for i in range(300)]}}
    fx = synth.load_fixtures()
    d4 = D.Driver(device=0, hostemu=True)
    c4 = D.Client(d4)
    for t in synth.psp_templates(fx):
        c4.AddTemplate(t)
    for k in synth.psp_constraints():
        c4.AddConstraint(k)
    r4 = d4.QueryMatching(D.TARGET_NAME, list(c4.constraints.values()), D.AugmentedUnstructured(D.Unstructured(huge), None, "Original"), tracing=True)
    assert "host evaluator" in r4.trace
