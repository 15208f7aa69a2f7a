"""Pins the oracle's Rego driver restatement against the reference's golden vectors."""
import pytest

import reference_tables as T
from conftest import gconst, ydocs
from oracle import target as tg
from oracle.client import AUDIT_EP, GATOR_EP, Client, ClientError
from oracle.gator import VerifyError, gator_test, verify_case
from oracle.rego_builtins import go_sprintf
from oracle.values import RSet, from_json, go_float_v, to_string

GT = "test/gator/test/fixtures/"
PSP = "pkg/webhook/testdata/psp-all-violations/"


def F(fx, *names):
    out = []
    for n in names:
        out.extend(gconst(fx, n))
    return out


# ------------------------------------------------------------------ pkg/gator/test/test_test.go:86-330
def test_gator_basic_no_violation(fixtures):
    assert gator_test(F(fixtures, "TemplateAlwaysValidate", "ConstraintAlwaysValidate", "Object")) == []


def test_gator_basic_violation(fixtures):
    res = gator_test(F(fixtures, "TemplateNeverValidate", "ConstraintNeverValidate", "Object"))
    # template, constraint and object are all reviewed (test_test.go:103-131)
    assert [r.msg for r, _ in res] == [T.MSG_NEVER_VALIDATE] * 3
    assert all(r.enforcement_action == "deny" and r.scoped_enforcement_actions is None for r, _ in res)


def test_gator_referential(fixtures):
    res = gator_test(F(fixtures, "TemplateReferential", "ConstraintReferential", "ObjectReferentialInventory",
                       "ObjectReferentialDeny"))
    assert sorted(r.msg for r, _ in res) == sorted(T.MSG_REFERENTIAL)
    assert gator_test(F(fixtures, "TemplateReferential", "ConstraintReferential", "ObjectReferentialInventory",
                        "ObjectReferentialAllow")) == []


def test_gator_misc(fixtures):
    assert gator_test([]) == []
    assert gator_test(F(fixtures, "ObjectReferentialInventory", "ObjectReferentialAllow")) == []
    assert gator_test(F(fixtures, "TemplateReferential")) == []
    with pytest.raises(ClientError):
        gator_test(F(fixtures, "ConstraintReferential"))


def test_gator_enforcement_points(fixtures):
    res = gator_test(F(fixtures, "TemplateNeverValidate", "ConstraintGatorValidate", "Object"))
    assert len(res) == 3
    for r, _ in res:
        assert (r.msg, r.enforcement_action, r.scoped_enforcement_actions) == ("never validate", "scoped", ["deny"])
    assert gator_test(F(fixtures, "TemplateNeverValidate", "ConstraintAuditValidate", "Object")) == []


def test_gator_twice(fixtures):
    res = gator_test(F(fixtures, "TemplateNeverValidateTwice", "ConstraintNeverValidateTwice", "Object"))
    assert sorted(r.msg for r, _ in res) == ["first message"] * 3 + ["second message"] * 3


def test_compile_error(fixtures):
    """pkg/gator/fixtures/fixtures.go:142-160 must surface from AddTemplate."""
    with pytest.raises(ClientError):
        Client().add_template(gconst(fixtures, "TemplateCompileError")[0])


# ------------------------------------------------------------------ pkg/gator/verify/runner_test.go
def V(fx, tmpl, cons, obj, inv=()):
    return verify_case(gconst(fx, tmpl)[0], gconst(fx, cons)[0], gconst(fx, obj)[0], [gconst(fx, i)[0] for i in inv])


def test_verify_namespaces(fixtures):
    # runner_test.go:840-967
    assert len(V(fixtures, "TemplateNeverValidate", "ConstraintExcludedNamespace", "ObjectIncluded")) == 1
    assert len(V(fixtures, "TemplateNeverValidate", "ConstraintExcludedNamespace", "ObjectExcluded")) == 0
    assert len(V(fixtures, "TemplateNeverValidate", "ConstraintIncludedNamespace", "ObjectIncluded")) == 1
    assert len(V(fixtures, "TemplateNeverValidate", "ConstraintIncludedNamespace", "ObjectExcluded")) == 0
    assert len(V(fixtures, "TemplateNeverValidate", "ConstraintClusterScope", "ObjectClusterScope")) == 1
    assert len(V(fixtures, "TemplateNeverValidate", "ConstraintClusterScope", "ObjectNamespaceScope")) == 0


def test_verify_namespace_selector(fixtures):
    # runner_test.go:968-1021
    a = ("TemplateNeverValidate", "ConstraintNamespaceSelector", "ObjectNamespaceScope")
    assert len(V(fixtures, *a, inv=["NamespaceSelected"])) == 1
    assert len(V(fixtures, *a, inv=["NamespaceNotSelected"])) == 0
    res = V(fixtures, *a)
    assert len(res) == 1 and "missing Namespace" in res[0].msg
    assert res[0].msg == ("unable to match constraints: error matching the requested object: object :failed to run "
                          "Match criteria: namespace selector for namespace-scoped object but missing Namespace")


def test_verify_admission_review(fixtures):
    # runner_test.go:1022-1175
    a = ("TemplateValidateUserInfo", "ConstraintAlwaysValidateUserInfo")
    assert V(fixtures, *a, "SystemAdmissionReview") == []
    res = V(fixtures, *a, "NonSystemAdmissionReview")
    assert [r.msg for r in res] == ["username is not allowed to perform this operation: foo"]
    assert V(fixtures, *a, "AdmissionReviewWithOldObject") == []
    for bad in ("InvalidAdmissionReview", "AdmissionReviewMissingRequest", "AdmissionReviewMissingObjectAndOldObject"):
        with pytest.raises(VerifyError):
            V(fixtures, *a, bad)
    with pytest.raises(tg.ReviewError):
        V(fixtures, *a, "DeleteAdmissionReviewWithNoOldObject")
    b = ("TemplateValidateUserInfo", "ConstraintAlwaysValidateUserInfoWithMatch")
    for bad in ("SystemAdmissionReviewMissingKind", "DeleteAdmissionReviewWithOldObjectMissingKind"):
        with pytest.raises(VerifyError):
            V(fixtures, *b, bad)


def test_verify_suite_yaml(fixtures):
    """test/gator/verify/suite.yaml: template uses object.get + sprintf."""
    d = "test/gator/verify/"
    tmpl = ydocs(fixtures, d + "template.yaml")[0]
    cons = ydocs(fixtures, d + "constraint.yaml")[0]
    allow = ydocs(fixtures, d + "allow_foo.yaml")[0]
    deny = ydocs(fixtures, d + "deny_foo.yaml")[0]
    assert verify_case(tmpl, cons, allow) == []
    assert len(verify_case(tmpl, cons, deny)) >= 1
    scoped = ydocs(fixtures, d + "constraint_with_scopedEA.yaml")[0]
    noep = ydocs(fixtures, d + "constraint_with_scopedEA_without_gator_ep.yaml")[0]
    assert len(verify_case(tmpl, scoped, deny)) >= 1
    assert verify_case(tmpl, noep, deny) == []


# ------------------------------------------------------------------ bats / demo exact messages
def _all_docs(fx, *paths):
    out = []
    for p in paths:
        out.extend(ydocs(fx, p))
    return out


def test_config1_demo_basic(fixtures):
    """BASELINE.json configs[0]: demo/basic K8sRequiredLabels."""
    d = "demo/basic/"
    objs = _all_docs(fixtures, d + "templates/k8srequiredlabels_template.yaml", d + "constraints/all_ns_must_have_gatekeeper.yaml",
                     d + "bad/bad_ns.yaml", d + "good/good_ns.yaml")
    res = gator_test(objs)
    assert [(r.msg, o["metadata"]["name"]) for r, o in res] == [(T.MSG_REQUIRED_LABELS_GATEKEEPER, "bad-ns")]
    assert res[0][0].metadata == {"details": {"missing_labels": ["gatekeeper"]}}


def test_bats_probes(fixtures):
    res = gator_test(_all_docs(fixtures, GT + "manifests/with-policies/with-violations.yaml"))
    assert T.MSG_PROBES in [r.msg for r, _ in res]
    assert gator_test(_all_docs(fixtures, GT + "manifests/with-policies/no-violations.yaml")) == []
    res = gator_test(_all_docs(fixtures, GT + "manifests/with-policies/with-violations-rego-v1.yaml"))
    assert any(r.enforcement_action == "deny" for r, _ in res)


def test_bats_policies_dir(fixtures):
    pol = [p for p in fixtures["yaml"] if p.startswith(GT + "policies/default/")]
    objs = _all_docs(fixtures, *pol)
    res = gator_test(objs + _all_docs(fixtures, GT + "manifests/no-policies/with-violations.yaml"))
    assert T.MSG_PROBES in [r.msg for r, _ in res]
    assert gator_test(objs + _all_docs(fixtures, GT + "manifests/no-policies/no-violations.yaml")) == []
    ref = [p for p in fixtures["yaml"] if p.startswith(GT + "manifests/referential-data/")]
    res = gator_test(objs + _all_docs(fixtures, *ref))
    assert T.MSG_INGRESS in [r.msg for r, _ in res]


def test_bats_enforcement_action_foo(fixtures):
    """test.bats:203-216: unknown enforcementAction still reports the violation."""
    objs = _all_docs(fixtures, GT + "policies/default/template_k8srequiredprobes.yaml",
                     GT + "policies/enforcement_action/k8srequiredprobes/foo.yaml",
                     GT + "manifests/no-policies/with-violations.yaml")
    res = gator_test(objs)
    assert T.MSG_PROBES in [r.msg for r, _ in res]
    assert all(r.enforcement_action != "deny" for r, _ in res)


def test_bats_geo(fixtures):
    d = "test/gator/oci-artifacts/"
    objs = _all_docs(fixtures, d + "templates/template_k8srequiredlabels.yaml", d + "constraints/ns-must-have-geo.yaml",
                     GT + "manifests/no-policies/violating-ns.yaml")
    assert T.MSG_REQUIRED_LABELS_GEO in [r.msg for r, _ in gator_test(objs)]


def test_bats_defaults(fixtures):
    res = gator_test(_all_docs(fixtures, GT + "manifests/with-policies/with-violations-and-defaults.yaml"))
    assert len(res) == 1
    r = res[0][0]
    assert r.msg == "aRequiredMessage" and r.metadata == {"details": {"missing_labels": ["aRequiredLabel"]}}


def test_bats_autoreject(fixtures):
    """test.bats:301 (expansion itself is out of scope: review the resultant pod shape directly)."""
    docs = _all_docs(fixtures, GT + "manifests/expansion/expansion-w-ns-selector.yaml")
    tmpl = [d for d in docs if d["kind"] == "ConstraintTemplate"][0]
    cons = [d for d in docs if d["apiVersion"].startswith("constraints.gatekeeper.sh")][0]
    c = Client()
    c.add_template(tmpl)
    c.add_constraint(cons)
    pod = {"apiVersion": "v1", "kind": "Pod", "metadata": {"name": "nginx-deployment-pod", "namespace": "default"}}
    res = c.review(tg.AugmentedUnstructured(tg.Unstructured(pod), None, "Generated"), GATOR_EP)
    assert [r.msg for r in res] == [T.MSG_AUTOREJECT]


# ------------------------------------------------------------------ PSP benchmark fixtures (config #2 policies)
def test_psp_all_violations(fixtures):
    """pkg/webhook/policy_benchmark_test.go:264-271: 'All constraints are applicable and all requests are violating'
    -- each pod violates (at least) the constraint it was written for."""
    tmpls = [ydocs(fixtures, p)[0] for p in sorted(fixtures["yaml"]) if p.startswith(PSP + "psp-templates/")]
    cons = [ydocs(fixtures, p)[0] for p in sorted(fixtures["yaml"]) if p.startswith(PSP + "psp-constraints/")]
    pods = [ydocs(fixtures, p)[0] for p in sorted(fixtures["yaml"]) if p.startswith(PSP + "psp-pods/")]
    assert len(tmpls) == 5 and len(cons) == 5 and len(pods) == 5
    c = Client()
    for t_ in tmpls:
        c.add_template(t_)
    for k in cons:
        c.add_constraint(k)
    own = {"nginx-host-filesystem": "K8sPSPHostFilesystem", "nginx-host-namespace": "K8sPSPHostNamespace",
           "nginx-host-networking-ports": "K8sPSPHostNetworkingPorts", "nginx-privileged": "K8sPSPPrivilegedContainer",
           "nginx-volume-types": "K8sPSPVolumeTypes"}
    got = {}
    for p in pods:
        res = c.review(tg.AugmentedUnstructured(tg.Unstructured(p), None, "Original"), AUDIT_EP)
        got[p["metadata"]["name"]] = sorted((r.constraint["kind"], r.msg) for r in res)
        assert own[p["metadata"]["name"]] in [k for k, _ in got[p["metadata"]["name"]]]
    assert got["nginx-privileged"] == [("K8sPSPPrivilegedContainer",
                                        'Privileged container is not allowed: nginx, securityContext: {"privileged": true}')]
    assert got["nginx-host-namespace"] == [("K8sPSPHostNamespace", "Sharing the host namespace is not allowed: nginx-host-namespace")]
    assert ("K8sPSPVolumeTypes", 'One of the volume types {"emptyDir", "hostPath"} is not allowed, pod: nginx-volume-types. '
            'Allowed volume types: ["configMap", "emptyDir", "projected", "secret", "downwardAPI", '
            '"persistentVolumeClaim", "flexVolume"]') in got["nginx-volume-types"]


def test_agilebank_families(fixtures):
    """demo/agilebank: every bad_resource violates, every good_resource passes (template compile coverage)."""
    d = "demo/agilebank/"
    tmpls = [ydocs(fixtures, p)[0] for p in sorted(fixtures["yaml"]) if p.startswith(d + "templates/")]
    cons = [ydocs(fixtures, p)[0] for p in sorted(fixtures["yaml"]) if p.startswith(d + "constraints/")]
    c = Client()
    for t_ in tmpls:
        c.add_template(t_)
    for k in cons:
        c.add_constraint(k)
    c.add_data({"apiVersion": "v1", "kind": "Namespace", "metadata": {"name": "production"}})

    def review(o):
        return c.review(tg.AugmentedUnstructured(tg.Unstructured(o), None, "Original"), AUDIT_EP)

    for p in sorted(fixtures["yaml"]):
        if p.startswith(d + "good_resources/"):
            for o in ydocs(fixtures, p):
                assert review(o) == [], p
    msgs = {}
    for p in sorted(fixtures["yaml"]):
        if p.startswith(d + "bad_resources/") and "duplicate_service" not in p and "deployment_nolimit" not in p:
            for o in ydocs(fixtures, p):
                msgs[p] = sorted(r.msg for r in review(o))
                assert msgs[p], p
    assert "container <opa> has no resource limits" in msgs[d + "bad_resources/opa_no_limits.yaml"]
    assert any("cpu limit <300m> is higher than the maximum allowed of <200m>" in x
               for x in msgs[d + "bad_resources/opa_limits_too_high.yaml"])
    assert any("has an invalid image repo" in x for x in msgs[d + "bad_resources/opa_wrong_repo.yaml"])


# ------------------------------------------------------------------ formatting
def test_sprintf_formatting():
    assert go_sprintf("%v", [RSet(["b", "a"])]) == '{"a", "b"}'
    assert go_sprintf("%v", [RSet([])]) == "set()"
    assert go_sprintf("%v %v", [from_json({"b": 1, "a": [True, None]}), "s"]) == '{"a": [true, null], "b": 1} s'
    assert go_sprintf("<%v: %v>", ["k", 5]) == "<k: 5>"
    assert go_sprintf("%s/%d", ["a", 3]) == "a/3"
    assert go_sprintf("%d", ["x"]) == "%!d(string=x)"
    assert go_sprintf("%v", []) == "%!v(MISSING)"
    assert go_sprintf("x", [1]) == "x%!(EXTRA int=1)"
    assert go_float_v(0.5) == "0.5" and go_float_v(1e21) == "1e+21" and go_float_v(1.5e-7) == "1.5e-07"
    assert go_float_v(123456789.25) == "1.2345678925e+08" or go_float_v(123456789.25) == "123456789.25"
    assert to_string(from_json({"x": "a\"b"})) == '{"x": "a\\"b"}'


# ------------------------------------------------------------------ pkg/gator/verify/runner_integer_test.go (pinned in round 6)
INTEGER_FLAVOURS = [("templateV1Beta1Integer", "constraintV1Beta1Integer"), ("templateV1Beta1IntegerNonStructural", "constraintV1Beta1Integer"),
                    ("templateV1Integer", "constraintV1Integer")]


def test_runner_run_integer(fixtures):
    """TestRunner_Run_Integer (runner_integer_test.go:231-308): K8sReplicaLimits in three template flavours (structural v1beta1,
    non-structural v1beta1, v1); the Deployment with 3 replicas yields 0 violations, the one with 100 yields exactly 1 -- integers of
    the parameters (min_replicas / max_replicas) compared with an integer of the object, whatever the CRD schema says about them."""
    consts = fixtures["go_consts"]["pkg/gator/verify/runner_integer_test.go"]
    allow, deny = consts["objectIntegerAllowed"]["docs"][0], consts["objectIntegerDisallowed"]["docs"][0]
    for tname, cname in INTEGER_FLAVOURS:
        tmpl, cons = consts[tname]["docs"][0], consts[cname]["docs"][0]
        assert verify_case(tmpl, cons, allow) == []
        res = verify_case(tmpl, cons, deny)
        assert len(res) == 1 and res[0].msg.startswith("The provided number of replicas is not allowed for deployment: disallowed-deployment. Allowed ranges: ")
