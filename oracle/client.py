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
    """any error the frameworks client or its Rego driver hands back to the caller"""


class Result:
    """frameworks types.Result: Target, Msg, Metadata{"details"}, Constraint, EnforcementAction, ScopedEnforcementActions"""

    __slots__ = ("target", "msg", "constraint", "metadata", "enforcement_action", "scoped_enforcement_actions")

    def __init__(self, msg, constraint, details=None, enforcement_action="deny", scoped_actions=None,
                 target=t.TARGET_NAME):
        self.target, self.msg, self.constraint = target, msg, constraint
        self.metadata = {"details": {} if details is None else details}
        self.enforcement_action, self.scoped_enforcement_actions = enforcement_action, scoped_actions

    def key(self):
        """what two Results must share to be the same Result (order-free comparison of result lists)"""
        meta = self.constraint.get("metadata", {})
        scoped = tuple(self.scoped_enforcement_actions) if self.scoped_enforcement_actions else ()
        return (self.constraint.get("kind"), meta.get("name"), self.msg, repr(hk(from_json(self.metadata))), self.enforcement_action, scoped)

    def __repr__(self):
        return "Result(%r, %s/%s, %s)" % (self.msg, self.constraint.get("kind"), self.constraint.get("metadata", {}).get("name"), self.enforcement_action)


def _field(node, *path):
    """unstructured.NestedFieldNoCopy: (value, found); nil on the way is "not found", anything else that is no map an error"""
    for name in path:
        if node is None:
            return None, False
        if not isinstance(node, dict):
            raise TypeError("%r accessor error: %r is not a map" % (".".join(path), node))
        if name not in node:
            return None, False
        node = node[name]
    return node, True


def template_source(ct: dict):
    """(kind, target, rego, libs) of a ConstraintTemplate: the one target's `code[engine=Rego].source`, or, without one, the legacy
    `rego` / `libs` fields (website/docs/constrainttemplates.md:216-232, pkg/fakes/fixtures.go:33-42)."""
    if not isinstance(ct.get("spec"), dict):
        raise ClientError("invalid ConstraintTemplate: spec must be an object")
    try:
        kind, found = _field(ct, "spec", "crd", "spec", "names", "kind")
    except TypeError:
        kind, found = None, False
    if not found:
        raise ClientError("invalid ConstraintTemplate: missing spec.crd.spec.names.kind")
    target_list = ct["spec"].get("targets")
    if not target_list or len(target_list) != 1:
        raise ClientError("invalid ConstraintTemplate: expected exactly 1 target, got %d" % len(target_list or ()))
    (tgt,) = target_list
    chosen = None
    for entry in tgt.get("code") or ():      # (the last Rego entry wins, as the driver's loop leaves it)
        if entry.get("engine") == "Rego":
            chosen = entry.get("source") or {}
    if chosen is None or chosen.get("rego") is None:
        chosen = tgt
    return kind, tgt.get("target"), chosen.get("rego"), list(chosen.get("libs") or ())


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


_ACTIONS = frozenset(("deny", "dryrun", "warn", "scoped"))


def get_enforcement_action(c):
    """util.GetEnforcementAction (pkg/util/enforcement_action.go:132-151; rows: enforcement_action_test.go:113-165):
    NestedString(spec.enforcementAction) -- absent is "", which defaults to deny; present but no string (or a spec that is no map)
    is ErrInvalidSpecEnforcementAction; a string outside the four actions is classified "unrecognized"."""
    try:
        value, found = _field(c, "spec", "enforcementAction")
    except TypeError:
        found, value = True, None
    if found and not isinstance(value, str):
        raise ClientError("unable to parse spec.enforcementAction")
    action = value if found and value else "deny"
    return action if action in _ACTIONS else "unrecognized"


def _enforcement_point_enabled(scoped_action, ep):
    """enforcementPointEnabled (enforcement_action.go:167-174)"""
    for point in scoped_action.get("enforcementPoints") or ():
        if isinstance(point, dict) and (point.get("name") == ep or point.get("name") == ALL_EP):
            return True
    return False


def scoped_actions_for_ep(ep, c):
    """util.ScopedActionForEP (enforcement_action.go:153-165; rows: enforcement_action_test.go:235-385): the Action of every
    scopedEnforcementActions entry enabled for `ep`; the JSON round trip into []ScopedEnforcementAction fails for anything that
    is no list of objects (convertToScopedEnforcementActions :119-130)"""
    try:
        listed, found = _field(c, "spec", "scopedEnforcementActions")
    except TypeError:
        listed, found = None, False
    if not found or listed is None:
        return []
    if not isinstance(listed, list) or any(not isinstance(entry, dict) for entry in listed):
        raise ClientError("could not convert JSON to scopedEnforcementActions")
    return [entry.get("action") for entry in listed if _enforcement_point_enabled(entry, ep)]


def _walk_defaults(schema, value):
    """Structural-schema defaulting (k8s apiextensions-apiserver pkg/apiserver/schema/defaulting Default): a property that is absent
    and has a `default` gets a copy of it; then properties / additionalProperties / items are visited below the value.  What
    Client.AddConstraint does through the template's generated CRD (test/gator/test/test.bats:277-291)."""
    if not isinstance(schema, dict):
        return value
    if isinstance(value, list):
        item = schema.get("items")
        return [_walk_defaults(item, member) for member in value] if isinstance(item, dict) else value
    if not isinstance(value, dict):
        return value
    properties = schema.get("properties") or {}
    for prop, prop_schema in properties.items():
        if isinstance(prop_schema, dict) and "default" in prop_schema and prop not in value:
            value[prop] = copy.deepcopy(prop_schema["default"])
    additional = schema.get("additionalProperties")
    for member in value:
        below = properties.get(member, additional if member not in properties else None)
        if isinstance(below, dict):
            value[member] = _walk_defaults(below, value[member])
    return value


def _mentions_default(schema):
    if isinstance(schema, dict):
        return any("default" in k for k in schema) or any(_mentions_default(v) for v in schema.values())
    if isinstance(schema, list):
        return any(_mentions_default(v) for v in schema)
    return isinstance(schema, str) and "default" in schema


def apply_schema_defaults(ct, c):
    """the constraint as the driver sees it: spec.parameters defaulted by the template's openAPIV3Schema (the schema OF parameters)"""
    try:
        schema, found = _field(ct, "spec", "crd", "spec", "validation", "openAPIV3Schema")
    except TypeError:
        return c
    if not found or not isinstance(schema, dict):
        return c
    c = copy.deepcopy(c)
    if not isinstance(c.get("spec"), dict):
        if not _mentions_default(schema):
            return c
        c["spec"] = {}
    if "parameters" not in c["spec"]:
        if "default" not in schema:
            return c
        c["spec"]["parameters"] = copy.deepcopy(schema["default"])
    c["spec"]["parameters"] = _walk_defaults(schema, c["spec"]["parameters"])
    return c


def _kind_and_name(resource):
    return resource.get("kind", ""), (resource.get("metadata") or {}).get("name", "")


class Client:
    """Restated constraintclient.Client for the single K8sValidationTarget: a template registry keyed by the lower-cased kind, a
    constraint registry keyed by (kind, name) holding the defaulted constraint with its Matcher, and the target's Namespace cache."""

    def __init__(self, driver=None, enforcement_points=(WEBHOOK_EP, AUDIT_EP, GATOR_EP)):
        self.driver = RegoDriver() if driver is None else driver
        self.cache = t.NsCache()
        self.templates = {}
        self.constraints = {}
        self.enforcement_points = tuple(enforcement_points)

    # ---- templates
    def add_template(self, ct):
        """Client.AddTemplate: the name is the lower-cased kind (ErrInvalidConstraintTemplate otherwise), the one target must be this
        client's, then the driver compiles it"""
        kind, target_name, _rego, _libs = template_source(ct)
        registry_key = kind.lower()
        given = (ct.get("metadata") or {}).get("name", "")
        if given != registry_key:
            raise ClientError("the ConstraintTemplate's name must be the lowercase of kind: got %r for kind %r" % (given, kind))
        if target_name != t.TARGET_NAME:
            raise ClientError("unknown target %r" % target_name)
        self.driver.add_template(ct)
        self.templates[registry_key] = ct

    def remove_template(self, ct):
        """Client.RemoveTemplate: the template and every constraint of its kind go"""
        registry_key = template_source(ct)[0].lower()
        self.driver.remove_template(ct)
        self.templates.pop(registry_key, None)
        self.constraints = {key: entry for key, entry in self.constraints.items() if key[0].lower() != registry_key}

    # ---- constraints
    def add_constraint(self, c, validate=True):
        """Client.AddConstraint: ErrMissingConstraintTemplate without the template; the target handler's ValidateConstraint
        (target.go:185-219; validate=False lets tests of the Match layer's own error paths, match_test.go, install selectors the
        handler would refuse); CRD defaulting; then ToMatcher (target.go:246-261 -- not part of ValidateConstraint, always run)"""
        kind, name = _kind_and_name(c)
        template = self.templates.get(kind.lower())
        if template is None:
            raise ClientError("missing ConstraintTemplate: %s" % kind)
        try:
            if validate:
                t.validate_constraint(c)
            defaulted = apply_schema_defaults(template, c)
            self.constraints[(kind, name)] = (defaulted, t.to_matcher(defaulted, self.cache))
        except t.ReviewError as err:
            raise ClientError(str(err))

    def remove_constraint(self, c):
        self.constraints.pop(_kind_and_name(c), None)

    # ---- data
    def _processed(self, obj):
        return t.process_data(obj if isinstance(obj, t.Unstructured) else t.Unstructured(obj))

    def add_data(self, obj):
        """Client.AddData: ProcessData names the storage path; the target's cache and the driver's data.inventory both take it"""
        handled, path, data = self._processed(obj)
        if handled:
            self.cache.add(path, data)
            self.driver.add_data(t.TARGET_NAME, path, data)

    def remove_data(self, obj):
        handled, path, _data = self._processed(obj)
        if handled:
            self.cache.remove(path)
            self.driver.remove_data(t.TARGET_NAME, path)

    # ---- review
    def _enforced(self, c, enforcement_point):
        """(enforcement action, scoped actions | None) or None when a scoped constraint names no action for this point"""
        action = get_enforcement_action(c)
        if action != "scoped":
            return action, None
        at_this_point = scoped_actions_for_ep(enforcement_point, c)
        return (action, at_this_point) if at_this_point else None

    def review(self, obj, enforcement_point=AUDIT_EP, namespace=None):
        """Client.Review -> list[Result].  HandleReview decides whether the target takes the object at all; every constraint (in key
        order) enforced at the point is matched -- a Matcher error becomes that constraint's autoreject Result; the matching ones go
        to the driver in ONE Query, and its Results are stamped with their constraint's actions.  `namespace` is the
        reviews.Namespace(nsMap) option (namespaceObject)."""
        handled, review = t.handle_review(obj)
        if not handled:
            return []
        rejected, to_query, actions = [], [], {}
        for key in sorted(self.constraints):
            c, matcher = self.constraints[key]
            enforced = self._enforced(c, enforcement_point)
            if enforced is None:
                continue
            try:
                if matcher.match_review(review):
                    to_query.append(c)
                    actions[id(c)] = enforced
            except t.ReviewError as err:
                rejected.append(Result("unable to match constraints: %s" % err, c, {}, *enforced))
        if not to_query:
            return rejected
        answered = self.driver.query(t.TARGET_NAME, to_query, review, namespace)
        for result in answered:
            result.enforcement_action, result.scoped_enforcement_actions = actions[id(result.constraint)]
        return rejected + answered
