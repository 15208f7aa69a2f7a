"""Golden truth tables transcribed BY HAND from the reference's Go table tests (inputs and expected outputs only).

Each table cites the file:line range it restates.  These pin the oracle (tests/test_oracle_match.py) and, through
the oracle, the HIP path.  Objects are JSON dicts as unstructured.Unstructured would hold them.
"""

# ---------------------------------------------------------------------------------------------------------------
# pkg/wildcard/wildcard_test.go:14-85  (Wildcard.Matches)          rows: (wildcard, candidate, want)
WILDCARD_MATCHES = [
    ("kube-system", "kube-system", True),
    ("kube-system", "gatekeeper-system", False),
    ("kube-*", "kube-system", True),
    ("kube-*", "gatekeeper-system", False),
    ("*-system", "kube-system", True),
    ("*-system", "kube-public", False),
    ("kube-", "kube-system", False),
    ("*-kube-*", "test-kube-test", True),
    ("*-kube-*", "my-kub-controller", False),
    ("-kube-", "test-kube-test", False),
    ("*--*", "my--namespace", True),
    ("**", "my:namespace", True),
]

# pkg/wildcard/wildcard_test.go:108-179  (Wildcard.MatchesGenerateName)
WILDCARD_GENERATE_NAME = [
    ("kube-system", "kube-system", False),
    ("kube-system", "gatekeeper-system", False),
    ("kube-*", "kube-system", True),
    ("kube-*", "gatekeeper-system", False),
    ("*-system", "kube-system", False),
    ("*-system", "kube-public", False),
    ("kube-", "kube-system", False),
    ("*-kube-*", "test-kube-test", True),
    ("-kube-", "test-kube-test", False),
    ("*-kube-*", "test-dev-kube-dev-test", True),
    ("*-kube-*", "my-kub-controller", False),
    ("*-kube-*", "my-controller-manager", False),
]


# ---------------------------------------------------------------------------------------------------------------
def _obj(group, kind, namespace, name, labels=None, generate_name=None):
    """makeObject (match_test.go:686-696): SetGroupVersionKind with empty version -> apiVersion '<group>/'."""
    o = {"apiVersion": (group + "/") if group else "", "kind": kind, "metadata": {}}
    if namespace:
        o["metadata"]["namespace"] = namespace
    if name:
        o["metadata"]["name"] = name
    if generate_name:
        o["metadata"]["generateName"] = generate_name
    if labels:
        o["metadata"]["labels"] = labels
    return o


def _nsobj(name, labels=None):
    """makeNamespace (match_test.go:698-715)"""
    o = {"apiVersion": "v1", "kind": "Namespace", "metadata": {"name": name}}
    if labels:
        o["metadata"]["labels"] = labels
    return o


def _ns(name, labels=None):
    """&corev1.Namespace{ObjectMeta{Name, Labels}}"""
    m = {"name": name}
    if labels:
        m["labels"] = labels
    return {"metadata": m}


_GK = ("group", "kind")
_LBL = {"labelname": "labelvalue"}

# pkg/mutation/match/match_test.go:17-684 (TestMatch)
# rows: (name, object, match, namespace, source, want_match, want_err)
MATCH_CASES = [
    ("nil object", None, {"namespaceSelector": {}}, None, "", False, True),
    ("match empty group kinds", _obj(*_GK, "", "name"), {"kinds": [{"kinds": [], "apiGroups": []}]}, None, "Original", True, False),
    ("match empty kinds", _obj(*_GK, "", "name"), {"kinds": [{"kinds": [], "apiGroups": ["*"]}]}, None, "Original", True, False),
    ("don't match empty kinds in other group", _obj(*_GK, "", "name"), {"kinds": [{"kinds": [], "apiGroups": ["rbac"]}]}, None, "Original", False, False),
    ("match kind with wildcard", _obj(*_GK, "", "name"), {"kinds": [{"kinds": ["*"], "apiGroups": ["*"]}]}, None, "Original", True, False),
    ("match group and no kinds specified should match", _obj(*_GK, "", "name"),
     {"kinds": [{"kinds": ["notmatching", "neithermatching"], "apiGroups": ["*"]}, {"apiGroups": ["*"]}]}, None, "Original", True, False),
    ("match kind and no group specified should match", _obj(*_GK, "", "name"),
     {"kinds": [{"kinds": ["kind", "neithermatching"]}]}, None, "Original", True, False),
    ("match kind and group explicit", _obj(*_GK, "", "name"),
     {"kinds": [{"kinds": ["notmatching", "neithermatching"], "apiGroups": ["*"]}, {"kinds": ["notmatching", "kind"], "apiGroups": ["*"]}]},
     None, "Original", True, False),
    ("kind group doesn't match", _obj(*_GK, "", "name"),
     {"kinds": [{"kinds": ["notmatching", "neithermatching"], "apiGroups": ["*"]}, {"kinds": ["notmatching", "kind"], "apiGroups": ["*"]}]},
     None, "Original", True, False),
    ("kind group don't match", _obj(*_GK, "", "name"),
     {"kinds": [{"kinds": ["notmatching", "neithermatching"], "apiGroups": ["*"]}, {"kinds": ["notmatching", "kind"], "apiGroups": ["notmatchinggroup"]}]},
     None, "Original", False, False),
    ("namespace matches", _obj(*_GK, "namespace", "name"), {"namespaces": ["nonmatching", "namespace"]}, _ns("namespace"), "Original", True, False),
    ("is a matching Namespace", _nsobj("matching"), {"namespaces": ["matching"]}, None, "Original", True, False),
    ("is not a matching Namespace", _nsobj("non-matching"), {"namespaces": ["matching"]}, None, "Original", False, False),
    ("namespaces configured, but cluster scoped", _obj(*_GK, "", "name"), {"namespaces": ["nonmatching", "namespace"]}, None, "Original", True, False),
    ("namespace prefix matches", _obj(*_GK, "kube-system", "name"), {"namespaces": ["nonmatching", "kube-*"]}, _ns("kube-system"), "Original", True, False),
    ("namespace is not in the matches list", _obj(*_GK, "namespace2", "name"), {"namespaces": ["nonmatching", "notmatchingeither"]}, None, "Original", False, False),
    ("has namespace fails if cluster scoped", _obj(*_GK, "namespace", "name"), {"scope": "Cluster"}, None, "Original", False, False),
    ("has namespace succeeds if namespace scoped", _obj(*_GK, "namespace", "name"), {"scope": "Namespaced"}, None, "Original", True, False),
    ("has namespace succeeds if scope is typo", _obj(*_GK, "namespace", "name"), {"scope": "cluster"}, None, "Original", True, False),
    ("without namespace succeeds if cluster scoped", _obj(*_GK, "", "name"), {"scope": "Cluster"}, None, "Original", True, False),
    ("without namespace fails if namespace scoped", _obj(*_GK, "", "name"), {"scope": "Namespaced"}, None, "Original", False, False),
    ("is namespace succeeds if cluster scoped", _nsobj("foo"), {"scope": "Cluster"}, None, "Original", True, False),
    ("is namespace fails if namespace scoped", _nsobj("foo"), {"scope": "Namespaced"}, None, "Original", False, False),
    ("object's namespace is excluded", _obj(*_GK, "namespace", "name"), {"excludedNamespaces": ["namespace"]}, None, "Original", False, False),
    ("object is an excluded Namespace", _nsobj("excluded"), {"excludedNamespaces": ["excluded"]}, None, "Original", False, False),
    ("object is not an excluded Namespace", _nsobj("not-excluded"), {"excludedNamespaces": ["excluded"]}, None, "Original", True, False),
    ("a namespace is excluded, but object is cluster scoped", _obj(*_GK, "", "name"), {"excludedNamespaces": ["namespace"]}, None, "Original", True, False),
    ("namespace is excluded by wildcard match", _obj(*_GK, "kube-system", "name"), {"excludedNamespaces": ["kube-*"]}, _ns("kube-system"), "Original", False, False),
    ("label selector", _obj(*_GK, "", "name", _LBL), {"labelSelector": {"matchLabels": {"labelname": "labelvalue"}}}, None, "Original", True, False),
    ("invalid label selector", _obj(*_GK, "", "name", _LBL), {"labelSelector": {"matchExpressions": [{"operator": "Invalid"}]}}, None, "Original", False, True),
    ("label selector not matching", _obj(*_GK, "", "name", _LBL),
     {"labelSelector": {"matchLabels": {"labelname": "labelvalue", "labelnotmatching": "foo"}}}, None, "Original", False, False),
    ("namespace selector", _obj(*_GK, "", "name"), {"namespaceSelector": {"matchLabels": {"labelname": "labelvalue"}}}, _ns("foo", _LBL), "Original", True, False),
    ("invalid namespace selector", _obj(*_GK, "", "name"), {"namespaceSelector": {"matchExpressions": [{"operator": "Invalid"}]}}, _ns("foo", _LBL), "Original", False, True),
    ("namespace selector not matching", _obj(*_GK, "foo", "name"),
     {"namespaceSelector": {"matchLabels": {"labelname": "labelvalue", "foo": "bar"}}}, _ns("foo", _LBL), "Original", False, False),
    ("namespace selector not matching, but cluster scoped", _obj(*_GK, "", "name"),
     {"namespaceSelector": {"matchLabels": {"labelname": "labelvalue", "foo": "bar"}}}, None, "Original", True, False),
    ("namespace selector is applied to the object, if the object is a namespace", _nsobj("namespace", _LBL),
     {"namespaceSelector": {"matchLabels": {"labelname": "labelvalue"}}}, None, "Original", True, False),
    ("namespace selector is applied to the namespace, and does not match", _nsobj("namespace", _LBL),
     {"namespaceSelector": {"matchLabels": {"labelname": "badvalue"}}}, None, "Original", False, False),
    ("namespace selector error on missing Namespace", _obj(*_GK, "foo", "name"),
     {"namespaceSelector": {"matchLabels": {"labelname": "badvalue"}}}, None, "Original", False, True),
    ("match name", _obj(*_GK, "", "name-foo"), {"name": "name-foo"}, None, "Original", True, False),
    ("match wildcard name", _obj(*_GK, "", "name-foo"), {"name": "name-*"}, None, "Original", True, False),
    ("missing asterisk in name wildcard does not match", _obj(*_GK, "", "name-foo"), {"name": "name-"}, None, "Original", False, False),
    ("wrong name does not match", _obj(*_GK, "", "name-foo"), {"name": "name-bar"}, None, "Original", False, False),
    ("no match with correct name and wrong namespace", _obj(*_GK, "namespace", "name-foo"),
     {"name": "name-foo", "namespaces": ["other-namespace"]}, None, "Original", False, False),
    ("match with same sources", _obj(*_GK, "namespace", "name-foo"),
     {"name": "name-foo", "namespaces": ["my-ns"], "source": "Generated"}, _ns("my-ns"), "Generated", True, False),
    ("match with empty source field on match obj", _obj(*_GK, "namespace", "name-foo"),
     {"name": "name-foo", "namespaces": ["my-ns"]}, _ns("my-ns"), "Generated", True, False),
    ("different source fields do not match", _obj(*_GK, "namespace", "name-foo"),
     {"name": "name-foo", "namespaces": ["my-ns"], "source": "Original"}, _ns("my-ns"), "Generated", False, False),
    ("empty source field on Matchable produces error", _obj(*_GK, "namespace", "name-foo"),
     {"name": "name-foo", "namespaces": ["my-ns"], "source": "Original"}, _ns("my-ns"), "", False, True),
]

# pkg/mutation/match/match_test.go:847-1040 (Test_namesMatch): (name, match.name, object, want)
_POD = ("*", "Pod")
NAMES_MATCH_CASES = [
    ("match name with wild card", "foo*", _obj(*_POD, "my-ns", "foo-bar"), True),
    ("match generate name with wild card", "foo*", _obj(*_POD, "my-ns", "", generate_name="foo-bar-"), True),
    ("match different name with wild card", "foo*", _obj(*_POD, "my-ns", "fob"), False),
    ("match different generate name with wild card", "foo*", _obj(*_POD, "my-ns", "", generate_name="fob-bar-"), False),
    ("match whole name with generate name", "foo", _obj(*_POD, "my-ns", "", generate_name="foo"), False),
    ("match prefix wildcard with generate name", "*foo", _obj(*_POD, "my-ns", "", generate_name="foo"), False),
    ("match later half of the name with wild card with generate name", "*-bar*", _obj(*_POD, "my-ns", "", generate_name="fob-bar"), True),
]


# ---------------------------------------------------------------------------------------------------------------
# pkg/target/target_integration_test.go:163-413 (TestConstraintEnforcement): 26 scenarios, each reviewed in 3
# shapes (AugmentedReview with Object, with OldObject only, AugmentedUnstructured) at :457-527.
# rows: (name, obj, ns|None, spec.match|None, allowed)
def _res(name, labels=None):
    """makeResource(GVK{Group:"some", Kind:"Thing"}, name, labels) -- version is empty (apiVersion 'some/')."""
    o = {"apiVersion": "some/", "kind": "Thing", "metadata": {"name": name}}
    if labels:
        o["metadata"]["labels"] = labels
    return o


def _typed_ns(name, labels=None):
    o = {"apiVersion": "v1", "kind": "Namespace", "metadata": {"name": name}}
    if labels:
        o["metadata"]["labels"] = labels
    return o


_KINDS = [{"apiGroups": ["some"], "kinds": ["Thing"]}]
_OL, _NL = {"obj": "label"}, {"ns": "label"}


def _everything(**over):
    mt = {"kinds": _KINDS, "namespaces": ["my-ns"], "labelSelector": {"matchLabels": {"obj": "label"}},
          "namespaceSelector": {"matchLabels": {"ns": "label"}}}
    mt.update(over)
    return mt


ENFORCEMENT_CASES = [
    ("match deny all", _res("foo"), _typed_ns("my-ns"), None, False),
    ("match namespace", _res("foo"), _typed_ns("my-ns"), {"namespaces": ["my-ns"]}, False),
    ("no match namespace", _res("foo"), _typed_ns("my-ns"), {"namespaces": ["not-my-ns"]}, True),
    ("match excludedNamespaces", _res("foo"), _typed_ns("my-ns"), {"excludedNamespaces": ["my-ns"]}, True),
    ("no match excludedNamespaces", _res("foo"), _typed_ns("my-ns"), {"excludedNamespaces": ["not-my-ns"]}, False),
    ("match labelselector", _res("foo", {"a": "label"}), _typed_ns("my-ns"), {"labelSelector": {"matchLabels": {"a": "label"}}}, False),
    ("no match labelselector", _res("foo", {"a": "label"}), _typed_ns("my-ns"), {"labelSelector": {"matchLabels": {"different": "label"}}}, True),
    ("match nsselector", _res("foo"), _typed_ns("my-ns", {"a": "label"}), {"namespaceSelector": {"matchLabels": {"a": "label"}}}, False),
    ("no match nsselector", _res("foo"), _typed_ns("my-ns", {"a": "label"}), {"namespaceSelector": {"matchLabels": {"different": "label"}}}, True),
    ("match kinds", _res("foo"), _typed_ns("my-ns"), {"kinds": _KINDS}, False),
    ("no match kinds", _res("foo"), _typed_ns("my-ns"), {"kinds": [{"apiGroups": ["different"], "kinds": ["Thing"]}]}, True),
    ("match name", _res("foo"), _typed_ns("my-ns"), {"name": "foo"}, False),
    ("no match name", _res("foo"), _typed_ns("my-ns"), {"name": "other-name"}, True),
    ("match name wildcard", _res("test-resource"), _typed_ns("my-ns"), {"name": "test-*"}, False),
    ("match everything", _res("foo", _OL), _typed_ns("my-ns", _NL), _everything(), False),
    ("match everything with scope as wildcard", _res("foo", _OL), _typed_ns("my-ns", _NL), _everything(scope="*"), False),
    ("match everything with scope as namespaced", _res("foo", _OL), _typed_ns("my-ns", _NL), _everything(scope="Namespaced"), False),
    ("match everything with scope as cluster", _res("foo", _OL), _typed_ns("my-ns", _NL), _everything(scope="Cluster"), True),
    ("match everything but kind", _res("foo", _OL), _typed_ns("my-ns", _NL),
     _everything(kinds=[{"apiGroups": ["different"], "kinds": ["Thing"]}]), True),
    ("match everything but namespace", _res("foo", _OL), _typed_ns("my-ns", _NL), _everything(namespaces=["different-ns"]), True),
    ("match everything but labelselector", _res("foo", _OL), _typed_ns("my-ns", _NL),
     _everything(labelSelector={"matchLabels": {"obj": "different-label"}}), True),
    ("match everything but nsselector", _res("foo", _OL), _typed_ns("my-ns", _NL),
     _everything(namespaceSelector={"matchLabels": {"ns": "different-label"}}), True),
    ("match everything cluster scoped", _res("foo", _OL), None, _everything(), False),
    ("match everything cluster scoped wildcard as scope", _res("foo", _OL), None, _everything(scope="*"), False),
    ("do not match everything cluster scoped namespaced as scope", _res("foo", _OL), None, _everything(scope="Namespaced"), True),
    ("match everything cluster scoped with cluster as scope", _res("foo", _OL), None, _everything(scope="Cluster"), False),
]

# ---------------------------------------------------------------------------------------------------------------
# Exact message strings the reference pins
MSG_REQUIRED_LABELS_GATEKEEPER = 'you must provide labels: {"gatekeeper"}'   # website/docs/constrainttemplates.md:118
MSG_REQUIRED_LABELS_GEO = 'you must provide labels: {"geo"}'                 # test/gator/test/test.bats:241,259
MSG_PROBES = "Container <tomcat> in your <Pod> <test-pod1> has no <readinessProbe>"   # test/gator/test/test.bats:80
MSG_INGRESS = "ingress host conflicts with an existing ingress <example-host.example.com>"   # test.bats:222
MSG_AUTOREJECT = ("unable to match constraints: error matching the requested object: nginx-deployment-pod "
                  ":failed to run Match criteria: namespace selector for namespace-scoped object but missing "
                  "Namespace")                                                # test/gator/test/test.bats:301
MSG_NEVER_VALIDATE = "never validate"                                        # pkg/gator/test/test_test.go:103-131
MSG_REFERENTIAL = ["same selector as service <gatekeeper-test-service-disallowed> in namespace <default>",
                   "same selector as service <gatekeeper-test-service-example> in namespace <default>"]  # :135-158

# pkg/audit/manager_test.go:231-273 (Test_truncateString): (str, size, want)
TRUNCATE_CASES = [("Hello world!", 12, "Hello world!"), ("Hello world!", 5, "He..."), ("Hello, world!", 0, "...")]

# pkg/audit/manager_test.go:41-103 (Test_SVQueue / Test_LimitQueue): three violations; descending pop order
SVQ_ITEMS = [
    {"group": "rbac.authorization.k8s.io", "version": "v1", "kind": "ClusterRoleBinding"},
    {"group": "authorization.k8s.io", "version": "v1", "kind": "SubjectAccessReview"},
    {"group": "rbac.authorization.k8s.io", "version": "v1", "kind": "RoleBinding"},
]
SVQ_POP_ORDER = [2, 0, 1]          # sv3, sv1, sv2
LIMITQ2_REMAINING = [0, 1]         # limit 2 keeps sv1, sv2 (pops sv1 then sv2)


# pkg/webhook/policy_test.go:1303-1339 (mixed results -> 2 deny, 2 warn) and :1395-1536 (count table), hand-transcribed:
# (name, [(msg, constraint name or None, enforcementAction, scopedEnforcementActions)], deny count, warn count)
PROCESS_RESULTS_CASES = [
    ("Only One Dry Run", [("test", "c", "dryrun", None)], 0, 0),
    ("Only One Deny", [("test", "c", "deny", None)], 1, 0),
    ("Only One Warn", [("test", "c", "warn", None)], 0, 1),
    ("One Dry Run and One Deny", [("test", "c", "dryrun", None), ("test", "c", "deny", None)], 1, 0),
    ("One Dry Run, One Deny, One Warn", [("test", "c", "dryrun", None), ("test", "c", "deny", None), ("test", "c", "warn", None)], 1, 1),
    ("Two Deny", [("test", "c", "deny", None), ("test", "c", "deny", None)], 2, 0),
    ("Two Warn", [("test", "c", "warn", None), ("test", "c", "warn", None)], 0, 2),
    ("Two Dry Run", [("test", "c", "dryrun", None), ("test", "c", "dryrun", None)], 0, 0),
    ("Random EnforcementAction", [("test", "c", "random", None)], 0, 0),
    ("export test mix (:1303-1339)", [None, ("missing constraint", None, "deny", None), ("deny", "deny", "deny", None),
                                      ("warn", "warn", "warn", None), ("dryrun", "dryrun", "dryrun", None),
                                      ("scoped", "scoped", "scoped", ["deny", "warn"]), ("invalid", "invalid", "invalid", None)], 2, 2),
]


# pkg/target/target_test.go:657-981 TestMatcher_Match, hand-transcribed (the "nil" row is an unhandled input type).
# (name, shape, request/object, review namespace, cached namespace, match, want matched, want error kind)
def _thing(name, namespace=None, labels=None, group="some", kind="Thing"):
    md = {"name": name}
    if namespace is not None:
        md["namespace"] = namespace
    if labels:
        md["labels"] = labels
    return {"apiVersion": group + "/" if group else "v1", "kind": kind, "metadata": md}


_FOO_MATCH = {"source": "All", "kinds": [{"kinds": ["Thing"], "apiGroups": ["some"]}], "scope": "Namespaced", "namespaces": ["my-ns"],
              "labelSelector": {"matchLabels": {"obj": "label"}}, "namespaceSelector": {"matchLabels": {"ns": "label"}}}
_NSSEL_MATCH = {"namespaceSelector": {"matchLabels": {"ns": "label"}}}
_MATCHED = _thing("bar", "foo", {"obj": "label"})
_UNMATCHED = _thing("bar", "foo", None, group="another", kind="thing")
_NAMESPACED_FOO = _thing("foo", "foo", {"obj": "label"})
_MY_NS = {"apiVersion": "v1", "kind": "Namespace", "metadata": {"name": "my-ns", "labels": {"ns": "label"}}}

MATCHER_MATCH_CASES = [
    ("AdmissionRequest supported", "request", {"object": _MATCHED}, None, None, _FOO_MATCH, False, None),
    ("unstructured.Unstructured supported", "object", _thing("foo"), None, None, _FOO_MATCH, False, None),
    ("Raw object doesn't unmarshal", "object", {"key": "Some invalid json"}, {"apiVersion": "v1", "kind": "Namespace", "metadata": {"name": "my-ns"}}, None, _FOO_MATCH, False, "request"),
    ("Match error", "request", {"object": _NAMESPACED_FOO}, None, None, _NSSEL_MATCH, False, "matching"),
    ("Success if Namespace not cached", "request", {"object": {"apiVersion": "v1", "kind": "Namespace", "metadata": {"name": "foo"}}}, None, None, _FOO_MATCH, False, None),
    ("AugmentedReview is supported", "request", {"object": _MATCHED}, _MY_NS, None, _FOO_MATCH, True, None),
    ("AugmentedUnstructured is supported", "object", _thing("foo", None, {"obj": "label"}), _MY_NS, None, _FOO_MATCH, True, None),
    ("Both object and old object are matched", "request", {"object": _MATCHED, "oldObject": _MATCHED}, _MY_NS, None, _FOO_MATCH, True, None),
    ("object is matched, old object is not matched", "request", {"object": _MATCHED, "oldObject": _UNMATCHED}, _MY_NS, None, _FOO_MATCH, True, None),
    ("object is not matched, old object is matched", "request", {"object": _UNMATCHED, "oldObject": _MATCHED}, _MY_NS, None, _FOO_MATCH, True, None),
    ("neither object is matched", "request", {"object": _UNMATCHED, "oldObject": _UNMATCHED}, _MY_NS, None, _FOO_MATCH, False, None),
    ("new object is not matched, old object is not specified", "request", {"object": _UNMATCHED}, _MY_NS, None, _FOO_MATCH, False, None),
    ("missing cached Namespace", "request", {"namespace": "foo", "object": _NAMESPACED_FOO}, None, None, _NSSEL_MATCH, False, "matching"),
    ("use cached Namespace no match", "request", {"namespace": "foo", "object": _NAMESPACED_FOO}, None,
     {"apiVersion": "v1", "kind": "Namespace", "metadata": {"name": "foo"}}, _NSSEL_MATCH, False, None),
    ("use cached Namespace match", "request", {"namespace": "foo", "object": _NAMESPACED_FOO}, None,
     {"apiVersion": "v1", "kind": "Namespace", "metadata": {"name": "foo", "labels": {"ns": "label"}}}, _NSSEL_MATCH, True, None),
    ("neither new or old object is specified", "request", {}, _MY_NS, None, _FOO_MATCH, False, "request"),
]
