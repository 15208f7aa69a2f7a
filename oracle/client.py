"""ORACLE (test infrastructure only -- never imported by the product path).

Restatement of the constraint-framework client + Rego driver the reference plugs into:
  github.com/open-policy-agent/frameworks/constraint  v0.0.0-20260616163050-e1eaa1bf6d62 (go.mod:18)
    pkg/client.Client            AddTemplate/AddConstraint/AddData/RemoveData/Review
    pkg/client/drivers/rego      Driver.Query
Both are third-party and absent from /root/reference; behaviour is restated from the reference's call sites and
tests:  pkg/gator/test/test.go:33-176 (call order), pkg/gator/test/test_test.go:86-330 (results, enforcement-point
scoping), pkg/target/target_integration_test.go:163-527, pkg/util/enforcement_action.go:138-174,
test/gator/test/test.bats:301 (autoreject message), pkg/drivers/k8scel/driver.go:162-251 (Query contract).
"""
from __future__ import annotations

import copy

from . import target as t
from .rego_interp import Interp, RegoEvalError
from .rego_parser import RegoSyntaxError
from .values import RObj, from_json, to_json, hk

WEBHOOK_EP = "validation.gatekeeper.sh"
AUDIT_EP = "audit.gatekeeper.sh"
GATOR_EP = "gator.gatekeeper.sh"
ALL_EP = "*"


class ClientError(Exception):
    pass


class Result:
    """frameworks types.Result"""

    def __init__(self, msg, constraint, details=None, enforcement_action="deny", scoped_actions=None,
                 target=t.TARGET_NAME):
        self.target = target
        self.msg = msg
        self.constraint = constraint
        self.metadata = {"details": details if details is not None else {}}
        self.enforcement_action = enforcement_action
        self.scoped_enforcement_actions = scoped_actions

    def key(self):
        c = self.constraint
        return (c.get("kind"), c.get("metadata", {}).get("name"), self.msg, repr(hk(from_json(self.metadata))),
                self.enforcement_action, tuple(self.scoped_enforcement_actions or ()))

    def __repr__(self):
        return "Result(%r, %s/%s, %s)" % (self.msg, self.constraint.get("kind"),
                                          self.constraint.get("metadata", {}).get("name"), self.enforcement_action)


def template_source(ct: dict):
    """(kind, target, rego, libs) of a ConstraintTemplate. `code[engine=Rego]` wins over legacy `rego`
    (website/docs/constrainttemplates.md:216-232, pkg/fakes/fixtures.go:33-42)."""
    spec = ct.get("spec")
    if not isinstance(spec, dict):
        raise ClientError("invalid ConstraintTemplate: spec must be an object")
    try:
        kind = spec["crd"]["spec"]["names"]["kind"]
    except (KeyError, TypeError):
        raise ClientError("invalid ConstraintTemplate: missing spec.crd.spec.names.kind")
    targets = spec.get("targets") or []
    if len(targets) != 1:
        raise ClientError("invalid ConstraintTemplate: expected exactly 1 target, got %d" % len(targets))
    tg = targets[0]
    rego, libs = None, []
    for code in tg.get("code") or []:
        if code.get("engine") == "Rego":
            src = code.get("source") or {}
            rego, libs = src.get("rego"), list(src.get("libs") or [])
    if rego is None:
        rego, libs = tg.get("rego"), list(tg.get("libs") or [])
    return kind, tg.get("target"), rego, libs


class RegoDriver:
    """Restated rego.Driver (Name()=="Rego")."""

    name = "Rego"

    def __init__(self):
        self.templates = {}      # lower(kind) -> Interp
        self.inventory = {}      # nested dict under data.inventory
        self._data = None        # its converted form (cache)

    def add_template(self, ct):
        kind, _, rego, libs = template_source(ct)
        if not rego:
            raise ClientError("template %s has no Rego source" % kind)
        try:
            ip = Interp([rego] + libs, data=None)
        except (RegoSyntaxError, RegoEvalError) as e:
            raise ClientError("invalid rego: %s" % e)
        if (ip.main_pkg, "violation") not in ip.rules:
            raise ClientError("invalid rego: missing required rule violation")
        for lib in ip.modules[1:]:
            if lib["package"][0] != "lib":
                raise ClientError("invalid rego: libs must be under package lib")
        self.templates[kind.lower()] = ip

    def remove_template(self, ct):
        kind, _, _, _ = template_source(ct)
        self.templates.pop(kind.lower(), None)

    def add_data(self, target, path, data):
        cur = self.inventory
        for p in path[:-1]:
            cur = cur.setdefault(p, {})
        cur[path[-1]] = copy.deepcopy(data)
        self._data = None

    def remove_data(self, target, path):
        cur = self.inventory
        for p in path[:-1]:
            cur = cur.get(p)
            if cur is None:
                return
        cur.pop(path[-1], None)
        self._data = None

    def query(self, target, constraints, review, namespace=None):
        """-> list[Result] (EnforcementAction left for the client to fill)."""
        review_json = t.review_input_json(review, namespace)
        if self._data is None:      # converted once per inventory state, not once per query
            self._data = from_json({"inventory": self.inventory})
        data = self._data
        out = []
        for c in constraints:
            ip = self.templates.get(c.get("kind", "").lower())
            if ip is None:
                raise ClientError("unknown constraint template: %s" % c.get("kind"))
            ip.data = data
            spec = c.get("spec") if isinstance(c.get("spec"), dict) else {}
            params = spec.get("parameters")
            inp = from_json({"review": review_json, "parameters": params if params is not None else {}})
            seen = set()
            for v in ip.violations(inp):
                if not isinstance(v, RObj) or not isinstance(v.get("msg"), str):
                    continue
                msg = v.get("msg")
                details = to_json(v.get("details")) if v.has("details") else {}
                k = (msg, repr(hk(from_json(details))))
                if k in seen:
                    continue
                seen.add(k)
                out.append(Result(msg, c, details))
        return out


def get_enforcement_action(c):
    """pkg/util/enforcement_action.go:132-151 (pinned by enforcement_action_test.go:113-165): default deny; anything
    outside {deny, dryrun, warn, scoped} is "unrecognized"; a spec / enforcementAction of the wrong type is an error."""
    spec = c.get("spec")
    if spec is None:
        return "deny"
    if not isinstance(spec, dict) or not isinstance(spec.get("enforcementAction", ""), str):
        raise ClientError("unable to parse spec.enforcementAction")   # ErrInvalidSpecEnforcementAction
    ea = spec.get("enforcementAction", "")
    if ea == "":
        return "deny"
    return ea if ea in ("deny", "dryrun", "warn", "scoped") else "unrecognized"


def scoped_actions_for_ep(ep, c):
    """pkg/util/enforcement_action.go:153-174 (pinned by enforcement_action_test.go:235-385): the actions whose
    enforcementPoints name `ep` or "*"; a scopedEnforcementActions value that is not a list of objects is an error."""
    spec = c.get("spec") if isinstance(c.get("spec"), dict) else {}
    seas = spec.get("scopedEnforcementActions")
    if seas is None:
        return []
    if not isinstance(seas, list) or not all(isinstance(x, dict) for x in seas):
        raise ClientError("could not convert JSON to scopedEnforcementActions")
    out = []
    for sea in seas:
        for p in sea.get("enforcementPoints") or []:
            if isinstance(p, dict) and p.get("name") in (ep, ALL_EP):
                out.append(sea.get("action"))
                break
    return out


def _default(schema, value):
    """Structural-schema defaulting (k8s apiextensions `default`), as Client.AddConstraint applies through the
    template's generated CRD (pinned by test/gator/test/test.bats:277-291)."""
    if not isinstance(schema, dict):
        return value
    if isinstance(value, dict):
        props = schema.get("properties") or {}
        for k, sub in props.items():
            if k not in value and isinstance(sub, dict) and "default" in sub:
                value[k] = copy.deepcopy(sub["default"])
            if k in value:
                value[k] = _default(sub, value[k])
        addl = schema.get("additionalProperties")
        if isinstance(addl, dict):
            for k in value:
                if k not in props:
                    value[k] = _default(addl, value[k])
    elif isinstance(value, list):
        items = schema.get("items")
        if isinstance(items, dict):
            value = [_default(items, v) for v in value]
    return value


def apply_schema_defaults(ct, c):
    try:
        schema = ct["spec"]["crd"]["spec"]["validation"]["openAPIV3Schema"]
    except (KeyError, TypeError):
        return c
    if not isinstance(schema, dict):
        return c
    c = copy.deepcopy(c)
    spec = c.get("spec")
    if not isinstance(spec, dict):
        if "default" not in str(schema):
            return c
        spec = c["spec"] = {}
    if "parameters" not in spec:
        if "default" in schema:
            spec["parameters"] = copy.deepcopy(schema["default"])
        else:
            return c
    spec["parameters"] = _default(schema, spec["parameters"])
    return c


class Client:
    """Restated constraintclient.Client for the single K8sValidationTarget."""

    def __init__(self, driver=None, enforcement_points=(WEBHOOK_EP, AUDIT_EP, GATOR_EP)):
        self.driver = driver or RegoDriver()
        self.cache = t.NsCache()
        self.templates = {}       # lower(kind) -> ct
        self.constraints = {}     # (kind, name) -> (constraint, matcher)
        self.enforcement_points = tuple(enforcement_points)

    # ---- state
    def add_template(self, ct):
        kind, target, _, _ = template_source(ct)
        name = (ct.get("metadata") or {}).get("name", "")
        if name != kind.lower():
            raise ClientError("the ConstraintTemplate's name must be the lowercase of kind: got %r for kind %r"
                              % (name, kind))
        if target != t.TARGET_NAME:
            raise ClientError("unknown target %r" % target)
        self.driver.add_template(ct)
        self.templates[kind.lower()] = ct

    def remove_template(self, ct):
        kind, _, _, _ = template_source(ct)
        self.driver.remove_template(ct)
        self.templates.pop(kind.lower(), None)
        for k in [k for k in self.constraints if k[0].lower() == kind.lower()]:
            del self.constraints[k]

    def add_constraint(self, c, validate=True):
        """validate: the target handler's ValidateConstraint (target.go:185-219), as frameworks' client does on
        AddConstraint; tests of the Match layer's own error paths (match_test.go) install invalid selectors with
        validate=False."""
        kind = c.get("kind", "")
        if kind.lower() not in self.templates:
            raise ClientError("missing ConstraintTemplate: %s" % kind)   # ErrMissingConstraintTemplate
        if validate:
            try:
                t.validate_constraint(c)
            except t.ReviewError as e:
                raise ClientError(str(e))
        name = (c.get("metadata") or {}).get("name", "")
        c = apply_schema_defaults(self.templates[kind.lower()], c)
        try:
            matcher = t.to_matcher(c, self.cache)      # (always: ToMatcher is not part of ValidateConstraint)
        except t.ReviewError as e:
            raise ClientError(str(e))
        self.constraints[(kind, name)] = (c, matcher)

    def remove_constraint(self, c):
        self.constraints.pop((c.get("kind", ""), (c.get("metadata") or {}).get("name", "")), None)

    def add_data(self, obj):
        handled, path, data = t.process_data(t.Unstructured(obj) if not isinstance(obj, t.Unstructured) else obj)
        if not handled:
            return
        self.cache.add(path, data)
        self.driver.add_data(t.TARGET_NAME, path, data)

    def remove_data(self, obj):
        handled, path, _ = t.process_data(t.Unstructured(obj) if not isinstance(obj, t.Unstructured) else obj)
        if handled:
            self.cache.remove(path)
            self.driver.remove_data(t.TARGET_NAME, path)

    # ---- review
    def review(self, obj, enforcement_point=AUDIT_EP, namespace=None):
        """-> list[Result]. `namespace` is the reviews.Namespace(nsMap) option (namespaceObject)."""
        handled, review = t.handle_review(obj)
        if not handled:
            return []
        matched = []
        results = []
        for key in sorted(self.constraints):
            c, matcher = self.constraints[key]
            ea = get_enforcement_action(c)
            scoped = None
            if ea == "scoped":
                scoped = scoped_actions_for_ep(enforcement_point, c)
                if not scoped:
                    continue
            try:
                ok = matcher.match_review(review)
            except t.ReviewError as e:
                results.append(Result("unable to match constraints: %s" % e, c, {}, ea, scoped))
                continue
            if ok:
                matched.append((c, ea, scoped))
        if matched:
            rs = self.driver.query(t.TARGET_NAME, [c for c, _, _ in matched], review, namespace)
            info = {id(c): (ea, scoped) for c, ea, scoped in matched}
            for r in rs:
                r.enforcement_action, r.scoped_enforcement_actions = info[id(r.constraint)]
                results.append(r)
        return results
