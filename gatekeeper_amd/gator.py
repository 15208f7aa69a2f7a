"""`gator test` on the MI355X engine: the reference's cluster-less harness loop and its three output formats.

  test()            pkg/gator/test/test.go:33-176   Test: templates, then constraints, then every object as data; then every
                                                    object (templates and constraints included) is reviewed at the gator
                                                    enforcement point with AugmentedUnstructured{Object, Namespace:
                                                    NamespaceForResource, Source: Original}.  Here ALL objects go through
                                                    ONE device launch (Client.ReviewBatch) instead of one Review per object.
                                                    Expansion (pkg/expansion) is out of scope: resultant resources are not
                                                    generated.
  results()         pkg/gator/test/types.go:54-77   GatorResponses.Results: sorted by enforcement action, then message
  format_output()   cmd/gator/test/test.go:140-245  "" (human friendly) | "json" | "yaml", --deny-only
  exit_code()       cmd/gator/test/test.go:131-138,247-266  1 when a result's action (or a scoped action) is `deny`

JSON / YAML field names: types.Result is third-party (frameworks/constraint pkg/types, absent from /root/reference); its
JSON tags are restated from the published source (target, msg, metadata, constraint, enforcementAction,
scopedEnforcementActions; all but target omitempty), GatorResult adds violatingObject and trace
(pkg/gator/test/types.go:11-22), the YAML form nests the framework result under `result` (an embedded struct without
an inline tag, go.yaml.in/yaml/v3) -- pinned by the reference only as far as test/gator/test/test.bats goes: valid JSON
(:158-166) and `.[i].result.msg` of the YAML (:27-50,168-175); anything beyond is "parity unpinned"."""
from __future__ import annotations

import copy
import json

from . import driver as D


class GatorError(Exception):
    """test.Test's error return (the Go error text, wrapped the same way)"""


class GatorResult:
    """pkg/gator/test/types.go:11-22: the framework Result plus the violating object (and the trace, never set here)"""

    def __init__(self, result, violating_object):
        self.result = result
        self.violating_object = violating_object
        self.trace = None

    # the fields the formats print, under the names of types.Result
    @property
    def msg(self):
        return self.result.msg

    @property
    def enforcement_action(self):
        return self.result.enforcement_action

    @property
    def scoped_enforcement_actions(self):
        return list(self.result.scoped_enforcement_actions or [])

    def framework_json(self):
        """types.Result as encoding/json writes it (omitempty on everything but target)"""
        r = self.result
        out = {"target": r.target}
        if r.msg:
            out["msg"] = r.msg
        if r.metadata:
            out["metadata"] = r.metadata
        if r.constraint is not None:
            out["constraint"] = r.constraint
        if r.enforcement_action:
            out["enforcementAction"] = r.enforcement_action
        if r.scoped_enforcement_actions:
            out["scopedEnforcementActions"] = list(r.scoped_enforcement_actions)
        return out


def _gvk(o):
    api = o.get("apiVersion", "") or ""
    group, _, version = api.rpartition("/")
    return group, version, o.get("kind", "") or ""


def is_template(o):     # pkg/gator/reader/read_resources.go:301-311
    g, _, k = _gvk(o)
    return g == "templates.gatekeeper.sh" and k == "ConstraintTemplate"


def is_constraint(o):   # pkg/gator/reader/read_resources.go:313-324
    return _gvk(o)[0] == "constraints.gatekeeper.sh"


def namespace_for_resource(objs, r):
    """pkg/gator/expand/expand.go:109-123: an EMPTY (non-nil) Namespace for cluster-scoped resources, the supplied Namespace
    object, a synthetic `default`, else nil"""
    rns = (r.get("metadata") or {}).get("namespace", "") or ""
    if rns == "":
        return {"metadata": {}}
    for o in objs:
        g, _, k = _gvk(o)
        if g == "" and k == "Namespace" and (o.get("metadata") or {}).get("name") == rns:
            return o
    if rns == "default":
        return {"metadata": {"name": "default"}}
    return None


def test(objs, client=None):
    """test.Test -> list[GatorResult] in the order of GatorResponses.Results().  Raises GatorError like the Go function
    returns an error (a bad template / constraint, data that cannot be added, a review the target handler rejects or
    that is beyond the engine's limits)."""
    c = client or D.Client(enforcement_points=(D.GATOR_EP,))
    # The reference expands every object through the ExpansionTemplates / mutators in the input and reviews the resultants
    # as well (pkg/gator/test/test.go:88-96,139-176, pkg/expansion).  Expansion is outside this engine's path (SURVEY.md
    # section 8: the mutation system is out of scope): input that asks for it is refused -- the caller falls back to the
    # reference's gator -- instead of silently yielding fewer results than the reference would.
    for o in objs:
        api, kind = str(o.get("apiVersion", "")), o.get("kind", "")
        if kind == "ExpansionTemplate" and api.startswith("expansion.gatekeeper.sh/"):
            raise GatorError("expansion unsupported: the input holds ExpansionTemplate %r (resultant resources are not generated by this engine)"
                             % ((o.get("metadata") or {}).get("name", "")))
        # (mutators alone change nothing: pkg/gator/expand applies them to the RESULTANTS of an ExpansionTemplate only
        #  (expand.go:69-107) -- without one the reference's results are those of the same input without the mutators)
    for o in objs:
        if is_template(o):
            try:
                c.AddTemplate(o)
            except (D.ClientError, D.EngineError) as e:
                raise GatorError("adding template %r: %s" % ((o.get("metadata") or {}).get("name", ""), e))
    for o in objs:
        if is_constraint(o):
            try:
                c.AddConstraint(o)
            except (D.ClientError, D.EngineError) as e:
                raise GatorError("adding constraint %r: %s" % ((o.get("metadata") or {}).get("name", ""), e))
    for o in objs:
        try:
            c.AddData(o)
        except (D.ClientError, D.EngineError) as e:
            g, v, k = _gvk(o)
            raise GatorError("adding data of GVK %r: %s" % ("%s/%s, Kind=%s" % (g, v, k) if g else "%s, Kind=%s" % (v, k), e))
    reviews = [D.AugmentedUnstructured(D.Unstructured(o), namespace_for_resource(objs, o), "Original") for o in objs]
    out = []
    for o, res in zip(objs, c.ReviewBatch(reviews, D.GATOR_EP) if reviews else []):
        if isinstance(res, Exception):
            g, v, k = _gvk(o)
            md = o.get("metadata") or {}
            raise GatorError("reviewing %s %s/%s: %s" % ("%s/%s, Kind=%s" % (g, v, k) if g else "%s, Kind=%s" % (v, k), md.get("namespace", "") or "",
                                                       md.get("name", "") or "", res))
        for r in res:
            r = copy.copy(r)
            r.constraint = copy.deepcopy(r.constraint)      # fromFrameworkResult: detached from the framework's references
            r.target = D.TARGET_NAME
            out.append(GatorResult(r, o))
    return results(out)


def results(rs):
    """GatorResponses.Results(): by enforcement action, then message.  Unpinned against the reference: Go's sort.Slice is
    not stable (results that tie on both keys may come out in another order there -- compare as multisets), and the
    reference's YAML output round-trips through JSON first (numbers become float64 / int64, nil metadata is dropped) while
    this module dumps the Python objects as they are."""
    return sorted(rs, key=lambda r: (r.enforcement_action or "", r.msg or ""))


def enforceable_failure(r):
    return r.enforcement_action == "deny" or "deny" in r.scoped_enforcement_actions


def exit_code(rs):
    return 1 if any(enforceable_failure(r) for r in rs) else 0


def _go_quote(s):
    """fmt's %q for the strings that occur here (strconv.Quote: JSON escaping is the same for printable ASCII / UTF-8)"""
    return json.dumps(s, ensure_ascii=False)


def format_output(all_results, fmt="", deny_only=False):
    rs = [r for r in all_results if not deny_only or enforceable_failure(r)]
    fmt = (fmt or "").lower()
    if fmt == "json":
        if not rs:
            return "null"       # json.MarshalIndent of a nil slice
        docs = []
        for r in rs:
            d = r.framework_json()           # embedded struct: its fields are inlined
            d["violatingObject"] = r.violating_object
            d["trace"] = r.trace
            docs.append(d)
        return json.dumps(docs, indent=4, ensure_ascii=False)
    if fmt == "yaml":
        import yaml
        docs = [{"result": _yaml_result(r), "violatingObject": copy.deepcopy(r.violating_object), "trace": r.trace} for r in rs]
        return yaml.safe_dump(docs, default_flow_style=False, sort_keys=False) if docs else "[]\n"
    buf = []
    for r in rs:
        o = r.violating_object
        md = o.get("metadata") or {}
        obj = "%s/%s %s" % (o.get("apiVersion", ""), o.get("kind", ""), md.get("name", ""))
        if md.get("namespace"):
            obj = "%s/%s %s/%s" % (o.get("apiVersion", ""), o.get("kind", ""), md["namespace"], md.get("name", ""))
        buf.append("%s: [%s] Message: %s\n" % (obj, _go_quote((r.result.constraint.get("metadata") or {}).get("name", "")), _go_quote(r.msg)))
        if r.trace is not None:
            buf.append("Trace: %s" % r.trace)
    return "".join(buf)


def _yaml_result(r):
    """types.Result through yaml.v3 without tags: lower-cased field names; the constraint (*unstructured.Unstructured) is a
    struct with one exported field, Object"""
    x = r.result
    return {"target": x.target, "msg": x.msg, "metadata": x.metadata, "constraint": {"object": copy.deepcopy(x.constraint)},
            "enforcementaction": x.enforcement_action, "scopedenforcementactions": list(x.scoped_enforcement_actions or [])}
