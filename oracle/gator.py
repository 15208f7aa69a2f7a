"""ORACLE (test infrastructure only -- never imported by the product path).

Restatement of the two `gator` harness loops the reference uses to drive the hot path without a cluster:
  pkg/gator/test/test.go:33-176      test.Test   (config #1 plumbing; expansion omitted -- pkg/expansion is out of scope)
  pkg/gator/verify/runner.go:323-437 runReview / validateAndReviewAdmissionReviewRequest
  pkg/gator/expand/expand.go:109-123 NamespaceForResource
  pkg/gator/reader/read_resources.go:301-324 IsTemplate / IsConstraint
"""
from __future__ import annotations

from . import match as m
from . import target as t
from .client import GATOR_EP, Client, ClientError


def is_template(o):
    g, _, k = m.obj_gvk(o)
    return g == "templates.gatekeeper.sh" and k == "ConstraintTemplate"


def is_constraint(o):
    return m.obj_gvk(o)[0] == "constraints.gatekeeper.sh"


def namespace_for_resource(objs, r):
    """expand.go:109-123 (returns an EMPTY non-nil Namespace for cluster-scoped resources)."""
    rns = m.obj_namespace(r)
    if rns == "":
        return {"metadata": {}}
    for o in objs:
        g, _, k = m.obj_gvk(o)
        if g == "" and k == "Namespace" and m.obj_name(o) == rns:
            return o
    if rns == "default":
        return {"metadata": {"name": "default"}}
    return None


def gator_test(objs, client=None):
    """test.Test -> list[(Result, violating obj)]"""
    c = client or Client(enforcement_points=(GATOR_EP,))
    for o in objs:
        if is_template(o):
            c.add_template(o)
    for o in objs:
        if is_constraint(o):
            c.add_constraint(o)
    for o in objs:
        c.add_data(o)
    out = []
    for o in objs:
        ns = namespace_for_resource(objs, o)
        au = t.AugmentedUnstructured(t.Unstructured(o), ns, "Original")
        for r in c.review(au, GATOR_EP):
            out.append((r, o))
    return out


class VerifyError(Exception):
    pass


def verify_case(template, constraint, obj, inventory=()):
    """runner.go:323-437 -> list[Result]; raises VerifyError/ClientError/ReviewError like the Go errors."""
    c = Client(enforcement_points=(GATOR_EP,))
    c.add_template(template)
    c.add_constraint(constraint)
    for inv in inventory:
        c.add_data(inv)
    g, _, k = m.obj_gvk(obj)
    if k == "AdmissionReview" and g == "admission.k8s.io":
        known = {"apiVersion", "kind", "request", "response"}
        if set(obj) - known:
            raise VerifyError("invalid admission review")                  # ErrInvalidK8sAdmissionReview
        req = obj.get("request")
        if req is None:
            raise VerifyError("missing admission request")                 # ErrMissingK8sAdmissionRequest
        if not isinstance(req.get("object"), dict) and not isinstance(req.get("oldObject"), dict):
            raise VerifyError("no object for review")                      # ErrNoObjectForReview
        for key in ("object", "oldObject"):
            o = req.get(key)
            if isinstance(o, dict) and not o.get("kind"):
                raise VerifyError("unmarshal object: Object 'Kind' is missing")   # ErrUnmarshallObject
        ar = t.AugmentedReview(t.AdmissionRequest(req), None, "Original")
        return c.review(ar, GATOR_EP)
    au = t.AugmentedUnstructured(t.Unstructured(obj), None, "Original")
    return c.review(au, GATOR_EP)
