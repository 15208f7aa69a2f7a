"""ORACLE (test infrastructure only -- never imported by the product path).

CPU restatement of the reference Match layer:
  pkg/wildcard/wildcard.go:17-41          Wildcard.Matches / MatchesGenerateName
  pkg/mutation/match/match.go:32-268      Matches + the 8 top-level matchers, IsNamespace
  pkg/mutation/match/match_types.go:13-64 Match / Kinds JSON shape
  pkg/mutation/types/mutator.go:14-46     source enum
Label selectors follow k8s.io/apimachinery v0.36.3 (go.mod:44, third-party, source absent):
  metav1.LabelSelectorAsSelector + labels.Requirement.Matches, restated from the published API semantics
   and pinned by pkg/mutation/match/match_test.go:370-555.

Objects are plain JSON dicts (what unstructured.Unstructured wraps).
"""
from __future__ import annotations

import re

ERR_MATCH = "failed to run Match criteria"          # match.go:16 ErrMatch


class MatchError(Exception):
    """(false, err) return of match.Matches; str() is Go's wrapped error text."""


# ------------------------------------------------------------------ pkg/wildcard/wildcard.go
def wildcard_matches(w: str, candidate: str) -> bool:
    """wildcard.go:17-30"""
    if w.startswith("*") and w.endswith("*"):
        inner = w[1:] if w.startswith("*") else w
        inner = inner[:-1] if inner.endswith("*") else inner
        return inner in candidate
    if w.startswith("*"):
        return candidate.endswith(w[1:])
    if w.endswith("*"):
        return candidate.startswith(w[:-1])
    return w == candidate


def wildcard_matches_generate_name(w: str, candidate: str) -> bool:
    """wildcard.go:32-41"""
    if w.startswith("*") and w.endswith("*"):
        inner = w[1:]
        inner = inner[:-1] if inner.endswith("*") else inner
        return inner in candidate
    if w.endswith("*"):
        return candidate.startswith(w[:-1])
    return False


# ------------------------------------------------------------------ object accessors (unstructured.Unstructured)
def _meta(obj):
    m = obj.get("metadata")
    return m if isinstance(m, dict) else {}


def _s(v):
    return v if isinstance(v, str) else ""


def obj_name(obj):
    return _s(_meta(obj).get("name"))


def obj_generate_name(obj):
    return _s(_meta(obj).get("generateName"))


def obj_namespace(obj):
    return _s(_meta(obj).get("namespace"))


def obj_labels(obj):
    """unstructured GetLabels -> NestedStringMap: any non-string value makes the whole map read as empty."""
    l = _meta(obj).get("labels")
    if not isinstance(l, dict):
        return {}
    if any(not isinstance(v, str) for v in l.values()):
        return {}
    return dict(l)


def parse_group_version(api_version: str):
    """schema.ParseGroupVersion"""
    if api_version == "" or api_version == "/":
        return "", ""
    n = api_version.count("/")
    if n == 0:
        return "", api_version
    if n == 1:
        i = api_version.index("/")
        return api_version[:i], api_version[i + 1:]
    return "", ""   # error in Go; GroupVersionKind() then yields empty group/version


def obj_gvk(obj):
    g, v = parse_group_version(_s(obj.get("apiVersion")))
    return g, v, _s(obj.get("kind"))


def is_namespace(obj) -> bool:
    """match.go:255-258"""
    g, _, k = obj_gvk(obj)
    return k == "Namespace" and g == ""


# ------------------------------------------------------------------ label selectors (apimachinery)
_QNAME = re.compile(r"^([A-Za-z0-9][-A-Za-z0-9_.]*)?[A-Za-z0-9]$")
_DNS1123_SUB = re.compile(r"^[a-z0-9]([-a-z0-9]*[a-z0-9])?(\.[a-z0-9]([-a-z0-9]*[a-z0-9])?)*$")
_LVAL = re.compile(r"^(([A-Za-z0-9][-A-Za-z0-9_.]*)?[A-Za-z0-9])?$")


def _valid_label_key(k: str) -> bool:
    parts = k.split("/")
    if len(parts) == 1:
        name = parts[0]
    elif len(parts) == 2:
        prefix, name = parts
        if prefix == "" or len(prefix) > 253 or not _DNS1123_SUB.match(prefix):
            return False
    else:
        return False
    return name != "" and len(name) <= 63 and bool(_QNAME.match(name))


def _valid_label_value(v: str) -> bool:
    return len(v) <= 63 and bool(_LVAL.match(v))


def selector_requirements(sel):
    """metav1.LabelSelectorAsSelector -> list of (key, op, values) or raises MatchError.
    Returns None for the empty selector (labels.Everything())."""
    ml = sel.get("matchLabels") or {}
    me = sel.get("matchExpressions") or []
    if len(ml) + len(me) == 0:
        return None
    reqs = []
    for k in sorted(ml):
        v = ml[k]
        reqs.append(_new_requirement(k, "Equals", [v]))
    for e in me:
        op = e.get("operator", "")
        if op not in ("In", "NotIn", "Exists", "DoesNotExist"):
            raise MatchError('"%s" is not a valid label selector operator' % op)
        reqs.append(_new_requirement(e.get("key", ""), op, list(e.get("values") or [])))
    return reqs


def _new_requirement(key, op, vals):
    """labels.NewRequirement validation."""
    if not isinstance(key, str) or not _valid_label_key(key):
        raise MatchError('key: Invalid value: "%s": name part must be non-empty' % key)
    if op in ("In", "NotIn"):
        if len(vals) == 0:
            raise MatchError("values: Invalid value: []: for 'in', 'notin' operators, values set can't be empty")
    elif op == "Equals":
        if len(vals) != 1:
            raise MatchError("values: Invalid value: exact-match compatibility requires one single value")
    else:
        if len(vals) != 0:
            raise MatchError("values: Invalid value: values set must be empty for exists and does not exist")
    for v in vals:
        if not isinstance(v, str) or not _valid_label_value(v):
            raise MatchError('values[0][%s]: Invalid value: "%s"' % (key, v))
    return key, op, vals


def selector_matches(reqs, labels) -> bool:
    if reqs is None:
        return True
    for key, op, vals in reqs:
        has = key in labels
        if op in ("In", "Equals"):
            ok = has and labels[key] in vals
        elif op == "NotIn":
            ok = (not has) or labels[key] not in vals
        elif op == "Exists":
            ok = has
        else:
            ok = not has
        if not ok:
            return False
    return True


# ------------------------------------------------------------------ the 8 top-level matchers (match.go)
VALID_SOURCES = ("All", "Generated", "Original")    # mutator.go:14-26


def kinds_match(match, obj, ns, source):
    """match.go:181-201"""
    kinds = match.get("kinds") or []
    if len(kinds) == 0:
        return True
    g, _, k = obj_gvk(obj)
    for kk in kinds:
        ks = kk.get("kinds") or []
        gs = kk.get("apiGroups") or []
        if not (len(ks) == 0 or "*" in ks or k in ks):
            continue
        if len(gs) == 0 or "*" in gs or g in gs:
            return True
    return False


def scope_match(match, obj, ns, source):
    """match.go:214-227"""
    has_ns = obj_namespace(obj) != "" or ns is not None
    is_ns = is_namespace(obj)
    scope = match.get("scope", "")
    if scope == "Cluster":
        return is_ns or not has_ns
    if scope == "Namespaced":
        return (not is_ns) and has_ns
    return True


def _effective_ns_name(obj, ns):
    if is_namespace(obj):
        return obj_name(obj)
    if ns is not None:
        return _s(_meta(ns).get("name"))
    if obj_namespace(obj) != "":
        return obj_namespace(obj)
    return None


def namespaces_match(match, obj, ns, source):
    """match.go:150-179"""
    nss = match.get("namespaces") or []
    if len(nss) == 0:
        return True
    name = _effective_ns_name(obj, ns)
    if name is None:
        return True
    return any(wildcard_matches(n, name) for n in nss)


def excluded_namespaces_match(match, obj, ns, source):
    """match.go:118-148"""
    nss = match.get("excludedNamespaces") or []
    if len(nss) == 0:
        return True
    name = _effective_ns_name(obj, ns)
    if name is None:
        return True
    return not any(wildcard_matches(n, name) for n in nss)


def label_selector_match(match, obj, ns, source):
    """match.go:103-116"""
    sel = match.get("labelSelector")
    if sel is None:
        return True
    reqs = selector_requirements(sel)
    return selector_matches(reqs, obj_labels(obj))


def namespace_selector_match(match, obj, ns, source):
    """match.go:73-101"""
    sel = match.get("namespaceSelector")
    if sel is None:
        return True
    is_ns = is_namespace(obj)
    if not is_ns and ns is None and obj_namespace(obj) == "":
        return True
    reqs = selector_requirements(sel)
    if is_ns:
        return selector_matches(reqs, obj_labels(obj))
    if ns is None:
        raise MatchError("namespace selector for namespace-scoped object but missing Namespace")
    return selector_matches(reqs, obj_labels(ns))


def names_match(match, obj, ns, source):
    """match.go:203-212"""
    name = match.get("name", "")
    if name == "":
        return True
    return wildcard_matches(name, obj_name(obj)) or wildcard_matches_generate_name(name, obj_generate_name(obj))


def source_match(match, obj, ns, source):
    """match.go:229-253"""
    m_src = match.get("source", "")
    t_src = source or ""
    if m_src == "":
        m_src = "All"
    elif m_src not in VALID_SOURCES:
        raise MatchError('invalid source field "%s"' % m_src)
    if t_src == "" and m_src != "All":
        raise MatchError("source field not specified for resource %s" % obj_name(obj))
    if m_src == "All":
        return True
    if t_src not in VALID_SOURCES:
        raise MatchError('invalid source field "%s"' % t_src)
    return m_src == t_src


TOP_LEVEL = (kinds_match, scope_match, namespaces_match, excluded_namespaces_match, label_selector_match,
             namespace_selector_match, names_match, source_match)


def matches(match: dict, obj, ns=None, source="") -> bool:
    """match.Matches (match.go:32-65). Raises MatchError('failed to run Match criteria: ...') on error."""
    if obj is None:
        raise MatchError("%s: obj must be non-nil" % ERR_MATCH)
    for fn in TOP_LEVEL:
        try:
            ok = fn(match, obj, ns, source)
        except MatchError as e:
            raise MatchError("%s: %s" % (ERR_MATCH, e))
        if not ok:
            return False
    return True
