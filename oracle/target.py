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
    """pkg/target/review.go:9-14: AdmissionRequest, Namespace, Source, IsAdmission"""
    __slots__ = ("admission_request", "namespace", "source", "is_admission")

    def __init__(self, admission_request, namespace=None, source="", is_admission=False):
        self.admission_request, self.namespace, self.source, self.is_admission = admission_request, namespace, source, is_admission


class AugmentedUnstructured:
    """pkg/target/data.go:26-31: Object, Namespace, Source, Operation"""
    __slots__ = ("object", "namespace", "source", "operation")

    def __init__(self, obj, namespace=None, source="", operation=""):
        self.object, self.namespace, self.source, self.operation = obj, namespace, source, operation


class GkReview:
    """pkg/target/review.go:16-21: the embedded AdmissionRequest plus unexported namespace / source / isAdmission"""
    __slots__ = ("request", "namespace", "source", "is_admission")

    def __init__(self, request, namespace=None, source="", is_admission=False):
        self.request, self.namespace, self.source, self.is_admission = request, namespace, source, is_admission


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
    if isinstance(obj, AugmentedReview):     # (value or pointer: one case each in the Go switch)
        review = GkReview(AdmissionRequest(copy.copy(obj.admission_request)), namespace=obj.namespace, source=obj.source, is_admission=obj.is_admission)
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


def _to_namespace(obj):
    """toNamespace (ns_cache.go:79-87): runtime.DefaultUnstructuredConverter.FromUnstructured into the typed corev1.Namespace -- a
    field of the wrong JSON type fails (third-party converter: restated for the struct-typed top-level fields and ObjectMeta's two
    map[string]string fields)"""
    def refuse(what):
        raise ReviewError("cannot cache type: cannot cache Namespace: " + what)
    for struct_field in ("metadata", "spec", "status"):
        value = obj.get(struct_field)
        if not (value is None or isinstance(value, dict)):
            refuse("%s must be an object" % struct_field)
    object_meta = obj.get("metadata") or {}
    for string_map in ("labels", "annotations"):
        value = object_meta.get(string_map)
        if value is None:
            continue
        if not isinstance(value, dict) or any(not isinstance(member, str) for member in value.values()):
            refuse("metadata.%s must be a map of strings" % string_map)


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
        _to_namespace(obj)
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


def _go_type(v):
    return {dict: "map[string]interface {}", list: "[]interface {}", str: "string", bool: "bool", int: "int64", float: "float64"}.get(type(v), type(v).__name__)


def nested_map(obj, *fields):
    """unstructured.NestedMap -> (map | None, found): nil on the way or an absent key is "not found"; a non-map on the way, or a
    non-map VALUE, is the accessor error (apimachinery unstructured/helpers.go NestedFieldNoCopy / jsonPath)"""
    value = obj
    for depth, name in enumerate(fields):
        if value is None:
            return None, False
        if not isinstance(value, dict):
            raise ReviewError(".%s accessor error: %r is of the type %s, expected map[string]interface{}" % (".".join(fields[:depth + 1]), value, _go_type(value)))
        if name not in value:
            return None, False
        value = value[name]
    if value is not None and not isinstance(value, dict):
        raise ReviewError(".%s accessor error: %r is of the type %s, expected map[string]interface{}" % (".".join(fields), value, _go_type(value)))
    return value, True


def convert_to_label_selector(selector):
    """convertToLabelSelector (target.go:221-231): the JSON round trip into metav1.LabelSelector {matchLabels map[string]string,
    matchExpressions []{key string, operator string, values []string}} -- a member of another JSON type fails json.Unmarshal"""
    def refuse(why):
        raise ReviewError("Could not convert JSON to LabelSelector: " + why)
    labels = selector.get("matchLabels")
    if labels is not None:
        if not isinstance(labels, dict) or any(not isinstance(k, str) or not isinstance(v, str) for k, v in labels.items()):
            refuse("matchLabels must be a map of strings")
    requirements = selector.get("matchExpressions")
    if requirements is None:
        return selector
    if not isinstance(requirements, list) or any(not isinstance(r, dict) for r in requirements):
        refuse("matchExpressions must be a list of requirements")
    for r in requirements:
        values = r.get("values")
        strings_ok = isinstance(r.get("key", ""), str) and isinstance(r.get("operator", ""), str)
        values_ok = values is None or (isinstance(values, list) and all(isinstance(v, str) for v in values))
        if not (strings_ok and values_ok):
            refuse("malformed requirement")
    return selector


def validate_constraint(constraint: dict):
    """K8sValidationTarget.ValidateConstraint (target.go:185-219; rows: target_test.go:42-399): for spec.match.labelSelector, then
    spec.match.namespaceSelector -- NestedMap, convertToLabelSelector, apimachinery's ValidateLabelSelector (third-party; the rules
    of labels.NewRequirement, restated in match.selector_requirements) under the field path spec.labelSelector for BOTH (as the
    reference passes it).  Raises ReviewError."""
    for which in ("labelSelector", "namespaceSelector"):
        selector, found = nested_map(constraint, "spec", "match", which)
        if not found or selector is None:
            continue
        typed = convert_to_label_selector(selector)
        try:
            m.selector_requirements(typed)
        except m.MatchError as err:
            raise ReviewError("spec.labelSelector: %s" % err)


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
    """K8sValidationTarget.ToMatcher (target.go:246-261; rows: target_test.go:562-655 TestToMatcher): NestedMap(spec.match) -- absent
    or null is the match-everything Matcher; an accessor error or a spec.match whose members do not convert into match.Match
    (convertToMatch) is ErrCreatingMatcher."""
    try:
        match_map, found = nested_map(constraint, "spec", "match")
    except ReviewError as err:
        raise ReviewError("%s: %s" % (ERR_CREATING_MATCHER, err))
    if not found or match_map is None:
        return Matcher(None, cache)
    why_not = _shape_error(match_map, _MATCH_SHAPE, "spec.match")
    if why_not:
        raise ReviewError("%s: %s" % (ERR_CREATING_MATCHER, why_not))
    return Matcher(match_map, cache)


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
