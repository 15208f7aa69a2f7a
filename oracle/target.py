"""ORACLE (test infrastructure only -- never imported by the product path).

CPU restatement of pkg/target (the K8sValidationTarget handler):
  pkg/target/target.go:81-179    HandleReview / handleReview, augmentedUnstructuredToAdmissionRequest,
                                 unstructuredToAdmissionRequest
  pkg/target/target.go:269-287   setObjectOnDelete (ErrOldObjectIsNil)
  pkg/target/target.go:40-79     ProcessData -> inventory path
  pkg/target/target.go:246-261   ToMatcher
  pkg/target/matcher.go:21-93    Matcher.Match, matchAny, gkReviewToObject
  pkg/target/ns_cache.go:15-87   nsCache
  pkg/target/review.go:9-29      AugmentedReview / gkReview
  pkg/target/data.go:26-31       AugmentedUnstructured

Go types are modelled as small Python classes; JSON objects are plain dicts.
"""
from __future__ import annotations

import copy

from . import match as m

TARGET_NAME = "admission.k8s.gatekeeper.sh"

ERR_REQUEST_OBJECT = "invalid request object"
ERR_MATCHING = "error matching the requested object"
ERR_OLD_OBJECT_IS_NIL = "oldObject cannot be nil for DELETE operations"


class ReviewError(Exception):
    pass


class AdmissionRequest(dict):
    """admissionv1.AdmissionRequest as its JSON dict: kind{group,version,kind}, name, namespace, operation,
    userInfo, object, oldObject, ... `object`/`oldObject` absent or None == RawExtension{Raw:nil}."""


class Unstructured(dict):
    """unstructured.Unstructured (value or pointer)."""


class AugmentedReview:
    """pkg/target/review.go:9-14"""

    def __init__(self, admission_request, namespace=None, source="", is_admission=False):
        self.admission_request = admission_request
        self.namespace = namespace
        self.source = source
        self.is_admission = is_admission


class AugmentedUnstructured:
    """pkg/target/data.go:26-31"""

    def __init__(self, obj, namespace=None, source="", operation=""):
        self.object = obj
        self.namespace = namespace
        self.source = source
        self.operation = operation


class GkReview:
    """pkg/target/review.go:16-21: the embedded AdmissionRequest plus unexported namespace/source/isAdmission."""

    def __init__(self, request, namespace=None, source="", is_admission=False):
        self.request = request
        self.namespace = namespace
        self.source = source
        self.is_admission = is_admission


def _raw(req, key):
    v = req.get(key)
    return v if isinstance(v, dict) else None


def unstructured_to_admission_request(obj) -> GkReview:
    """target.go:159-179"""
    g, v, k = m.obj_gvk(obj)
    req = AdmissionRequest({
        "kind": {"group": g, "version": v, "kind": k},
        "object": copy.deepcopy(dict(obj)),
        "name": m.obj_name(obj),
        "namespace": m.obj_namespace(obj),
    })
    return GkReview(req)


def augmented_unstructured_to_admission_request(au: AugmentedUnstructured) -> GkReview:
    """target.go:140-157"""
    review = unstructured_to_admission_request(au.object)
    review.namespace = au.namespace
    if au.operation != "":
        review.request["operation"] = au.operation
    if au.operation == "DELETE":
        review.request["oldObject"] = review.request["object"]
        review.request["object"] = None
    review.source = au.source
    return review


def set_object_on_delete(review: GkReview):
    """target.go:269-287"""
    if review.request.get("operation", "") == "DELETE":
        if _raw(review.request, "oldObject") is None:
            raise ReviewError(ERR_OLD_OBJECT_IS_NIL)
        review.request["object"] = review.request["oldObject"]


def handle_review(obj):
    """target.go:81-138 -> (handled, GkReview|None). Raises ReviewError on error."""
    if isinstance(obj, AugmentedReview):
        review = GkReview(AdmissionRequest(copy.copy(obj.admission_request)), obj.namespace, obj.source,
                          obj.is_admission)
    elif isinstance(obj, AugmentedUnstructured):
        review = augmented_unstructured_to_admission_request(obj)
    elif isinstance(obj, AdmissionRequest):
        review = GkReview(AdmissionRequest(copy.copy(obj)))
    elif isinstance(obj, Unstructured):
        review = unstructured_to_admission_request(obj)
    else:
        return False, None
    set_object_on_delete(review)
    return True, review


def process_data(obj):
    """target.go:40-79 -> (handled, path, data). Raises ReviewError."""
    if not isinstance(obj, Unstructured):
        return False, None, None
    g, v, k = m.obj_gvk(obj)
    if v == "":
        raise ReviewError("%s: resource %s has no version" % (ERR_REQUEST_OBJECT, m.obj_name(obj)))
    if k == "":
        raise ReviewError("%s: resource %s has no kind" % (ERR_REQUEST_OBJECT, m.obj_name(obj)))
    gv = v if g == "" else g + "/" + v
    if m.obj_namespace(obj) == "":
        path = ["cluster", gv, k, m.obj_name(obj)]
    else:
        path = ["namespace", m.obj_namespace(obj), gv, k, m.obj_name(obj)]
    return True, path, dict(obj)


class NsCache:
    """ns_cache.go:15-87"""

    def __init__(self):
        self.cache = {}

    def add(self, key, obj):
        if not isinstance(obj, dict):
            raise ReviewError("cannot cache type")
        g, _, k = m.obj_gvk(obj)
        if not (g == "" and k == "Namespace"):
            return
        # toNamespace (ns_cache.go:79-87): conversion into the typed corev1.Namespace fails on fields of the wrong JSON
        # type (third-party converter; restated for the typed top-level fields and string maps of ObjectMeta)
        for f in ("metadata", "spec", "status"):
            if f in obj and obj[f] is not None and not isinstance(obj[f], dict):
                raise ReviewError("cannot cache type: cannot cache Namespace: %s must be an object" % f)
        md = obj.get("metadata") or {}
        for f in ("labels", "annotations"):
            v = md.get(f)
            if v is not None and not (isinstance(v, dict) and all(isinstance(x, str) for x in v.values())):
                raise ReviewError("cannot cache type: cannot cache Namespace: metadata.%s must be a map of strings" % f)
        self.cache["/".join(key)] = obj

    def remove(self, key):
        self.cache.pop("/".join(key), None)

    def get_namespace(self, name):
        return self.cache.get("/".join(["cluster", "v1", "Namespace", name]))


class Matcher:
    """matcher.go:15-71"""

    def __init__(self, match, cache: NsCache):
        self.match = match
        self.cache = cache

    def match_review(self, review) -> bool:
        if self.match is None:
            return True
        if not isinstance(review, GkReview):
            raise ReviewError("unexpected review format")
        obj = _raw(review.request, "object")
        old = _raw(review.request, "oldObject")
        # gkReviewToObject (matcher.go:73-93): Unstructured.UnmarshalJSON rejects a document without a `kind`
        # (apimachinery UnstructuredJSONScheme, third-party) => ErrRequestObject.  The echoed raw bytes are the caller's
        # encoding of the object; restated here as compact, key-sorted JSON ("parity unpinned" for that substring).
        for which, o in (("object", obj), ("oldObject", old)):
            if o is not None and not (isinstance(o, dict) and isinstance(o.get("kind"), str) and o.get("kind")):
                import json as _json
                raise ReviewError("%s: failed to unmarshal gkReview %s %s" % (
                    ERR_REQUEST_OBJECT, which, _json.dumps(o, separators=(",", ":"), sort_keys=True, ensure_ascii=False)))
        ns = review.namespace
        req_ns = review.request.get("namespace", "") or ""
        if ns is None and req_ns != "":
            ns = self.cache.get_namespace(req_ns)
        return self._match_any(ns, review.source, [obj, old])

    def _match_any(self, ns, source, objs) -> bool:
        nil_obj = 0
        for obj in objs:
            if obj is None:
                nil_obj += 1
                continue
            try:
                if m.matches(self.match, obj, ns, source):
                    return True
            except m.MatchError as e:
                raise ReviewError("%s: %s :%s" % (ERR_MATCHING, m.obj_name(obj), e))
        if nil_obj == len(objs):
            raise ReviewError("%s: neither object nor old object are defined" % ERR_REQUEST_OBJECT)
        return False


def validate_constraint(constraint: dict):
    """K8sValidationTarget.ValidateConstraint (target.go:185-219): spec.match.labelSelector / namespaceSelector must be
    maps (unstructured.NestedMap), decode into metav1.LabelSelector (matchLabels: map[string]string, matchExpressions:
    [{key, operator, values: []string}]) and pass apimachinery's ValidateLabelSelector (third-party; same rules as
    labels.NewRequirement, restated in match.selector_requirements).  Raises ReviewError.
    Pinned by target_test.go:42-399 (11 cases)."""
    spec = constraint.get("spec")
    mt = spec.get("match") if isinstance(spec, dict) else None
    if mt is None:
        return
    if not isinstance(mt, dict):
        raise ReviewError("spec.match accessor error: %r is of the type %s, expected map[string]interface{}" % (mt, type(mt).__name__))
    for field in ("labelSelector", "namespaceSelector"):
        if field not in mt or mt[field] is None:
            continue
        sel = mt[field]
        if not isinstance(sel, dict):
            raise ReviewError(".spec.match.%s accessor error: %r is of the type %s, expected map[string]interface{}" % (field, sel, type(sel).__name__))
        ml = sel.get("matchLabels")
        if ml is not None and not (isinstance(ml, dict) and all(isinstance(k, str) and isinstance(v, str) for k, v in ml.items())):
            raise ReviewError("Could not convert JSON to LabelSelector: matchLabels must be a map of strings")
        me = sel.get("matchExpressions")
        if me is not None:
            if not isinstance(me, list) or not all(isinstance(e, dict) for e in me):
                raise ReviewError("Could not convert JSON to LabelSelector: matchExpressions must be a list of requirements")
            for e in me:
                vals = e.get("values")
                if not isinstance(e.get("key", ""), str) or not isinstance(e.get("operator", ""), str) or not (
                        vals is None or (isinstance(vals, list) and all(isinstance(v, str) for v in vals))):
                    raise ReviewError("Could not convert JSON to LabelSelector: malformed requirement")
        try:
            m.selector_requirements(sel)
        except m.MatchError as e:
            raise ReviewError("spec.labelSelector: %s" % e)


ERR_CREATING_MATCHER = "unable to create matcher"   # target.go ErrCreatingMatcher

# The typed shape of spec.match (pkg/mutation/match/match.go:32-65 Match, :67-78 Kinds; metav1.LabelSelector / LabelSelectorRequirement)
# as runtime.DefaultUnstructuredConverter.FromUnstructured sees it (target.go:253 convertToMatch): a field of the wrong JSON type is
# an error, an unknown field is ignored, null is the zero value.
_SELECTOR = ("struct", {"matchLabels": ("map", "str"), "matchExpressions": ("list", ("struct", {"key": "str", "operator": "str", "values": ("list", "str")}))})
_MATCH_SHAPE = ("struct", {"source": "str", "scope": "str", "name": "str", "namespaces": ("list", "str"), "excludedNamespaces": ("list", "str"),
                           "kinds": ("list", ("struct", {"apiGroups": ("list", "str"), "kinds": ("list", "str")})),
                           "labelSelector": _SELECTOR, "namespaceSelector": _SELECTOR})


def _shape_error(v, shape, where):
    """None, or why `v` does not convert into the typed field"""
    if v is None:
        return None
    if shape == "str":
        return None if isinstance(v, str) else "%s: expected string, got %s" % (where, type(v).__name__)
    kind, inner = shape
    if kind == "list":
        if not isinstance(v, list):
            return "%s: expected list, got %s" % (where, type(v).__name__)
        for i, x in enumerate(v):
            e = _shape_error(x, inner, "%s[%d]" % (where, i))
            if e:
                return e
        return None
    if not isinstance(v, dict):
        return "%s: expected map, got %s" % (where, type(v).__name__)
    if kind == "map":
        for k, x in v.items():
            e = _shape_error(x, inner, "%s.%s" % (where, k))
            if e:
                return e
        return None
    for k, sub in inner.items():
        e = _shape_error(v.get(k), sub, "%s.%s" % (where, k))
        if e:
            return e
    return None


def to_matcher(constraint: dict, cache: NsCache) -> Matcher:
    """target.go:246-261 (pinned by target_test.go:562-655 TestToMatcher): spec.match absent => match-everything Matcher; a
    spec.match that is not a map, or whose fields do not convert into match.Match, is ErrCreatingMatcher."""
    spec = constraint.get("spec")
    mt = spec.get("match") if isinstance(spec, dict) else None
    if mt is None:
        return Matcher(None, cache)
    if not isinstance(mt, dict):
        raise ReviewError("%s: .spec.match accessor error: %r is of the type %s, expected map[string]interface{}" % (ERR_CREATING_MATCHER, mt, type(mt).__name__))
    err = _shape_error(mt, _MATCH_SHAPE, "spec.match")
    if err:
        raise ReviewError("%s: %s" % (ERR_CREATING_MATCHER, err))
    return Matcher(mt, cache)


def review_input_json(review: GkReview, namespace_obj=None) -> dict:
    """JSON encoding of the embedded admissionv1.AdmissionRequest as the Rego driver sees it (input.review).
    Follows the struct tags of k8s.io/api/admission/v1 AdmissionRequest (third-party): `omitempty` on
    subResource, requestKind, requestResource, requestSubResource, name, namespace, dryRun; RawExtension fields
    (object, oldObject, options) encode as null when empty; uid/kind/resource/operation/userInfo always present.
    namespaceObject: injected from reviews.Namespace(nsMap) (pkg/util/namespace.go:15-48) when provided."""
    r = review.request
    kind = r.get("kind") or {}
    res = r.get("resource") or {}
    out = {
        "uid": r.get("uid", ""),
        "kind": {"group": kind.get("group", ""), "version": kind.get("version", ""), "kind": kind.get("kind", "")},
        "resource": {"group": res.get("group", ""), "version": res.get("version", ""),
                     "resource": res.get("resource", "")},
        "operation": r.get("operation", ""),
        "userInfo": r.get("userInfo") or {},
        "object": _raw(r, "object"),
        "oldObject": _raw(r, "oldObject"),
        "options": r.get("options"),
    }
    for k in ("subResource", "requestSubResource", "name", "namespace"):
        if r.get(k):
            out[k] = r[k]
    for k in ("requestKind", "requestResource", "dryRun"):
        if r.get(k) is not None:
            out[k] = r[k]
    if namespace_obj is not None:
        out["namespaceObject"] = namespace_obj
    return out
