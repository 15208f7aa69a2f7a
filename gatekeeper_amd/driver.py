"""Host-side mirror of the reference's plugin interface for the hot path, on top of the C ABI (include/gkgpu.h).

The reference's Go toolchain is absent from this image, so the layer a Go maintainer would write with cgo
 is mirrored here in Python with the same names, argument meaning and error behaviour:

  Driver   -- drivers.Driver (frameworks constraint/pkg/client/drivers; compile-checked exemplar
              pkg/drivers/k8scel/driver.go:56-264): Name/AddTemplate/RemoveTemplate/AddConstraint/RemoveConstraint/
              AddData/RemoveData/Query/Dump/GetDescriptionForStat.  Name() == "Rego" so templates route here.
  Client   -- the slice of constraintclient.Client the callers use (pkg/gator/test/test.go:33-176,
              pkg/webhook/policy.go:826, pkg/audit/manager.go:621,719): AddTemplate/AddConstraint/AddData/Review,
              including enforcement-point scoping (pkg/util/enforcement_action.go:132-174), CRD parameter
              defaulting and the autoreject result for match errors.
  AdmissionRequest / Unstructured / AugmentedReview / AugmentedUnstructured -- pkg/target/review.go:9-29,
              pkg/target/data.go:26-31 input shapes of K8sValidationTarget.HandleReview (pkg/target/target.go:81-138).

All evaluation (Match + violation predicates) runs in the HIP kernels; this module only marshals JSON, and renders
messages for the sparse violating pairs through gk_render.  No CPU fallback exists.
"""
from __future__ import annotations

import copy
import ctypes as C
import json
import re

import numpy as np

from . import _lib as L

TARGET_NAME = "admission.k8s.gatekeeper.sh"
WEBHOOK_EP = "validation.gatekeeper.sh"
AUDIT_EP = "audit.gatekeeper.sh"
GATOR_EP = "gator.gatekeeper.sh"
ALL_EP = "*"

_SOURCES = {"": L.GK_SRC_EMPTY, "Original": L.GK_SRC_ORIGINAL, "Generated": L.GK_SRC_GENERATED, "All": L.GK_SRC_ALL}


class EngineError(Exception):
    def __init__(self, code, msg):
        super().__init__(msg)
        self.code = code


class UnsupportedError(EngineError):
    """GK_ERR_UNSUPPORTED: the template/constraint cannot run on the device plan (keep it on the CPU Rego driver)."""


class LimitError(EngineError):
    """GK_ERR_LIMIT: the review is beyond the engine's limits (an array that element predicates iterate has more than
    255 elements, or holds an OBJECT where the compiled predicates iterate array elements -- Rego's `x[_]` would walk the
    object's values).  The engine reports such reviews in `too_big` and NEVER answers "no violations" for them: every caller
    below fails closed (raises / returns this error for that review) so the object can go to the reference CPU driver."""

    def __init__(self, what="review"):
        super().__init__(L.GK_ERR_LIMIT, "%s is beyond the engine's limits (more than 255 elements in an array that "
                                         "constraint predicates iterate, or an object where they iterate array elements): "
                                         "not evaluated on the device" % what)


# ---------------------------------------------------------------------------------------------- review shapes
class AdmissionRequest(dict):
    """admissionv1.AdmissionRequest as its JSON dict."""


class Unstructured(dict):
    """unstructured.Unstructured"""


class AugmentedReview:
    def __init__(self, admission_request, namespace=None, source="", is_admission=False):
        self.admission_request = admission_request
        self.namespace = namespace
        self.source = source
        self.is_admission = is_admission


class AugmentedUnstructured:
    def __init__(self, obj, namespace=None, source="", operation=""):
        self.object = obj
        self.namespace = namespace
        self.source = source
        self.operation = operation


class ReviewIn:
    """One gk_review_in."""

    __slots__ = ("kind", "json", "namespace", "ns_object", "source", "operation")

    def __init__(self, kind, body, namespace=None, ns_object=None, source="", operation=""):
        self.kind = kind
        self.json = body if isinstance(body, (bytes, bytearray)) else json.dumps(body).encode()
        self.namespace = None if namespace is None else json.dumps(namespace).encode()
        self.ns_object = None if ns_object is None else json.dumps(ns_object).encode()
        self.source = _SOURCES.get(source, L.GK_SRC_INVALID)
        self.operation = operation.encode() if operation else None


def to_review_in(obj, ns_object=None):
    """HandleReview input shapes -> gk_review_in; returns None for unhandled types (handled=false)."""
    if isinstance(obj, AugmentedReview):
        return ReviewIn(L.GK_REVIEW_ADMISSION_REQUEST, dict(obj.admission_request), obj.namespace, ns_object, obj.source)
    if isinstance(obj, AugmentedUnstructured):
        return ReviewIn(L.GK_REVIEW_OBJECT, dict(obj.object), obj.namespace, ns_object, obj.source, obj.operation)
    if isinstance(obj, AdmissionRequest):
        return ReviewIn(L.GK_REVIEW_ADMISSION_REQUEST, dict(obj), None, ns_object)
    if isinstance(obj, Unstructured):
        return ReviewIn(L.GK_REVIEW_OBJECT, dict(obj), None, ns_object)
    if isinstance(obj, ReviewIn):
        return obj
    return None


# ---------------------------------------------------------------------------------------------- engine wrapper
class EvalResult:
    def __init__(self, lib, ptr):
        o = ptr.contents
        self.n_reviews, self.n_constraints, self.n_tiles = o.n_reviews, o.n_constraints, o.n_tiles
        nc, nt = self.n_constraints, self.n_tiles
        self.constraint_ids = np.ctypeslib.as_array(o.constraint_ids, (nc,)).copy() if nc else np.zeros(0, np.uint32)

        def bm(p):
            if not p or nc * nt == 0:
                return np.zeros((nc, nt), np.uint64)
            return np.ctypeslib.as_array(p, (nc * nt,)).reshape(nc, nt).copy()

        has = bool(o.viol)
        self.viol = bm(o.viol) if has else None
        self.err = bm(o.err) if has else None
        self.match = bm(o.match) if o.match else None
        self.too_big = np.ctypeslib.as_array(o.too_big, (nt,)).copy() if (o.too_big and nt) else np.zeros(nt, np.uint64)
        self.counts = np.ctypeslib.as_array(o.counts, (nc,)).copy() if (o.counts and nc) else np.zeros(nc, np.uint32)
        self.list = (np.ctypeslib.as_array(o.list, (o.list_len * 2,)).reshape(-1, 2).copy()
                     if (o.list and o.list_len) else np.zeros((0, 2), np.uint32))
        self.list_total = o.list_total
        self.n_overflow = o.n_overflow
        self.kernel_ms, self.fast_kernel_ms = o.kernel_ms, o.fast_kernel_ms
        self.algo_bytes, self.n_rows, self.n_launches = o.algo_bytes, o.n_rows, o.n_launches
        self.n_rows_read = o.n_rows_read
        self.algo_bytes_once, self.n_plan_groups = o.algo_bytes_once, o.n_plan_groups
        # reviews beyond the device's limits that the engine's host evaluator answered (their bits are in the bitmaps, not in too_big)
        self.host_evaluated = [int(o.host_evaluated[i]) for i in range(o.n_host_evaluated)] if o.n_host_evaluated else []
        self.lds_bytes = o.lds_bytes
        self.kernel_text_hash = int(o.kernel_text_hash)
        self.d_viol, self.d_err, self.d_counts = o.d_viol, o.d_err, o.d_counts
        lib.gk_eval_free(ptr)

    @staticmethod
    def bits(bitmap_row, n):
        """indices of set bits in one bitmap row"""
        b = np.unpackbits(bitmap_row.view(np.uint8), bitorder="little")[:n]
        return np.nonzero(b)[0]

    def too_big_reviews(self):
        """indices of the reviews the engine refused to evaluate (beyond its limits)"""
        return [int(r) for r in self.bits(self.too_big, self.n_reviews)]

    def pairs(self, which="viol"):
        """sorted list of (constraint_id, review) for the chosen bitmap"""
        bm = getattr(self, which)
        out = []
        for row in range(self.n_constraints):
            cid = int(self.constraint_ids[row])
            for r in self.bits(bm[row], self.n_reviews):
                out.append((cid, int(r)))
        return sorted(out)


_QNAME_RE = re.compile(r"^([A-Za-z0-9][-A-Za-z0-9_.]*)?[A-Za-z0-9]$")
_DNS1123_SUB_RE = re.compile(r"^[a-z0-9]([-a-z0-9]*[a-z0-9])?(\.[a-z0-9]([-a-z0-9]*[a-z0-9])?)*$")
_LVAL_RE = re.compile(r"^(([A-Za-z0-9][-A-Za-z0-9_.]*)?[A-Za-z0-9])?$")


def _valid_label_key(k):
    parts = k.split("/")
    if len(parts) == 1:
        name = parts[0]
    elif len(parts) == 2:
        prefix, name = parts
        if prefix == "" or len(prefix) > 253 or not _DNS1123_SUB_RE.match(prefix):
            return False
    else:
        return False
    return name != "" and len(name) <= 63 and bool(_QNAME_RE.match(name))


def check_matcher(constraint):
    """K8sValidationTarget.ToMatcher (pkg/target/target.go:246-261; target_test.go:562-655): spec.match must be a map whose fields
    convert into the typed match.Match (pkg/mutation/match/match.go:32-78) -- strings where it has strings, lists of strings,
    label selectors of the metav1 shape; unknown fields are ignored, null is the zero value.  Run by Client.AddConstraint
    whether or not the constraint is validated (ToMatcher is not part of ValidateConstraint).  Raises ClientError."""
    spec = constraint.get("spec")
    if spec is not None and not isinstance(spec, dict):   # NestedMap's accessor error on the way to spec.match
        raise ClientError("unable to create matcher: spec is %s, not a map" % type(spec).__name__)
    mt = spec.get("match") if spec else None
    if mt is None:
        return
    if not isinstance(mt, dict):
        raise ClientError("unable to create matcher: spec.match is %s, not a map" % type(mt).__name__)
    strings, string_lists, selectors = ("source", "scope", "name"), ("namespaces", "excludedNamespaces"), ("labelSelector", "namespaceSelector")

    def bad(what, want):
        raise ClientError("unable to create matcher: spec.match.%s must be %s" % (what, want))

    def str_list(v, what):
        if v is not None and not (isinstance(v, list) and all(x is None or isinstance(x, str) for x in v)):
            bad(what, "a list of strings")
    for f in strings:
        if mt.get(f) is not None and not isinstance(mt[f], str):
            bad(f, "a string")
    for f in string_lists:
        str_list(mt.get(f), f)
    kinds = mt.get("kinds")
    if kinds is not None:
        if not isinstance(kinds, list):
            bad("kinds", "a list")
        for i, k in enumerate(kinds):
            if k is None:
                continue
            if not isinstance(k, dict):
                bad("kinds[%d]" % i, "a map")
            str_list(k.get("apiGroups"), "kinds[%d].apiGroups" % i)
            str_list(k.get("kinds"), "kinds[%d].kinds" % i)
    for f in selectors:
        sel = mt.get(f)
        if sel is None:
            continue
        if not isinstance(sel, dict):
            bad(f, "a map")
        ml = sel.get("matchLabels")
        if ml is not None and not (isinstance(ml, dict) and all(v is None or isinstance(v, str) for v in ml.values())):
            bad(f + ".matchLabels", "a map of strings")
        me = sel.get("matchExpressions")
        if me is not None:
            if not isinstance(me, list):
                bad(f + ".matchExpressions", "a list")
            for i, e in enumerate(me):
                if e is None:
                    continue
                if not isinstance(e, dict):
                    bad("%s.matchExpressions[%d]" % (f, i), "a map")
                for key in ("key", "operator"):
                    if e.get(key) is not None and not isinstance(e[key], str):
                        bad("%s.matchExpressions[%d].%s" % (f, i, key), "a string")
                str_list(e.get("values"), "%s.matchExpressions[%d].values" % (f, i))


def validate_constraint(constraint):
    """K8sValidationTarget.ValidateConstraint (pkg/target/target.go:185-219), run by Client.AddConstraint as the
    frameworks client does: spec.match.labelSelector / namespaceSelector must be maps that decode into a
    metav1.LabelSelector and pass apimachinery's ValidateLabelSelector.  Raises ClientError."""
    spec = constraint.get("spec")
    if spec is not None and not isinstance(spec, dict):   # unstructured.NestedMap(spec, match, labelSelector): accessor error
        raise ClientError("spec must be an object")
    mt = spec.get("match") if spec else None
    if mt is None:
        return
    if not isinstance(mt, dict):
        raise ClientError("spec.match must be an object")
    for field in ("labelSelector", "namespaceSelector"):
        sel = mt.get(field)
        if sel is None:
            continue
        if not isinstance(sel, dict):
            raise ClientError("spec.match.%s must be an object" % field)
        ml = sel.get("matchLabels")
        if ml is not None and not (isinstance(ml, dict) and all(isinstance(k, str) and isinstance(v, str) for k, v in ml.items())):
            raise ClientError("Could not convert JSON to LabelSelector: matchLabels must be a map of strings")
        for k, v in (ml or {}).items():
            if not _valid_label_key(k) or len(v) > 63 or not _LVAL_RE.match(v):
                raise ClientError("spec.labelSelector.matchLabels: Invalid value: %r" % ({k: v},))
        me = sel.get("matchExpressions")
        if me is None:
            continue
        if not isinstance(me, list) or not all(isinstance(e, dict) for e in me):
            raise ClientError("Could not convert JSON to LabelSelector: matchExpressions must be a list of requirements")
        for e in me:
            key, op, vals = e.get("key", ""), e.get("operator", ""), e.get("values")
            if not isinstance(key, str) or not isinstance(op, str) or not (vals is None or (isinstance(vals, list) and all(isinstance(v, str) for v in vals))):
                raise ClientError("Could not convert JSON to LabelSelector: malformed requirement")
            vals = vals or []
            if op not in ("In", "NotIn", "Exists", "DoesNotExist"):
                raise ClientError("spec.labelSelector.matchExpressions.operator: Invalid value: %r: not a valid selector operator" % op)
            if op in ("In", "NotIn") and not vals:
                raise ClientError("spec.labelSelector.matchExpressions.values: Required value: must be specified when `operator` is 'In' or 'NotIn'")
            if op in ("Exists", "DoesNotExist") and vals:
                raise ClientError("spec.labelSelector.matchExpressions.values: Forbidden: may not be specified when `operator` is 'Exists' or 'DoesNotExist'")
            if not _valid_label_key(key):
                raise ClientError("spec.labelSelector.matchExpressions.key: Invalid value: %r" % key)
            for v in vals:
                if len(v) > 63 or not _LVAL_RE.match(v):
                    raise ClientError("spec.labelSelector.matchExpressions.values: Invalid value: %r" % v)


def process_validation_results(results):
    """Row a12 -- pkg/webhook/policy.go:265-399: the deny / warn message lists the validating webhook builds from the hot
    path's results ("[<constraint name>] <msg>", :390,394).  Results without a constraint, with an unknown enforcement
    action, or scoped results without a valid scoped action are skipped (validatedEnforcementActions, :478-502)."""
    deny, warn = [], []
    for r in results:
        if r is None or r.constraint is None:
            continue
        if r.enforcement_action == "scoped":
            actions = [a for a in (r.scoped_enforcement_actions or []) if a in ("deny", "dryrun", "warn")]
            if not actions:
                continue
        elif r.enforcement_action in ("deny", "dryrun", "warn"):
            actions = [r.enforcement_action]
        else:
            continue
        name = (r.constraint.get("metadata") or {}).get("name", "")
        for a in actions:
            if a == "deny":
                deny.append("[%s] %s" % (name, r.msg))
            elif a == "warn":
                warn.append("[%s] %s" % (name, r.msg))
    return deny, warn


def truncate_string(s, size):
    """pkg/audit/manager.go:1039-1048: Go slices bytes."""
    b = s.encode("utf-8")
    if len(b) > size:
        if size > 3:
            size -= 3
        return b[:size].decode("utf-8", errors="surrogateescape") + "..."
    return s


def obj_gvk(obj):
    api = obj.get("apiVersion", "") if isinstance(obj.get("apiVersion"), str) else ""
    g, _, v = api.rpartition("/")
    return g, v, obj.get("kind", "") if isinstance(obj.get("kind"), str) else ""


def review_object(review):
    """the audited object of a review shape (Unstructured / AugmentedUnstructured)"""
    inner = getattr(review, "object", review)
    inner = getattr(inner, "object", inner)
    return inner if isinstance(inner, dict) else {}


class Table:
    def __init__(self, engine, handle, statuses, n):
        self.engine, self.handle, self.statuses, self.n = engine, handle, statuses, n

    def eval(self, want_match=False, download=True, want_list=False, collect_only=False, host_eval=True):
        """Launch + collect (or, with collect_only, just collect the pending launch()es).  host_eval=False: the device's answer alone
        (GK_EVAL_DEVICE_ONLY: reviews beyond the device's limits stay in too_big)."""
        lib = self.engine.lib
        flags = (L.GK_EVAL_WANT_MATCH if want_match else 0) | (0 if download else L.GK_EVAL_NO_DOWNLOAD) | (
            L.GK_EVAL_WANT_LIST if want_list else 0) | (L.GK_EVAL_COLLECT if collect_only else 0) | (0 if host_eval else L.GK_EVAL_DEVICE_ONLY)
        out = C.POINTER(L.gk_eval_out)()
        self.engine._check(lib.gk_table_eval(self.engine.handle, self.handle, flags, C.byref(out)))
        return EvalResult(lib, out)

    def launch(self, want_match=False, time_each=False, kernel_only=False):
        """Enqueue one evaluation launch without waiting (GK_EVAL_ASYNC); a later eval() collects.  time_each: an event pair around
        this launch alone (GK_EVAL_TIME_EACH: isolated kernel durations); kernel_only: the dominant kernel without the totals kernel
        behind it (GK_EVAL_KERNEL_ONLY: consecutive launches of that kernel under the collecting call's one event pair)"""
        out = C.POINTER(L.gk_eval_out)()
        flags = L.GK_EVAL_ASYNC | (L.GK_EVAL_WANT_MATCH if want_match else 0) | (L.GK_EVAL_TIME_EACH if time_each else 0) | (
            L.GK_EVAL_KERNEL_ONLY if kernel_only else 0)
        self.engine._check(self.engine.lib.gk_table_eval(self.engine.handle, self.handle, flags, C.byref(out)))

    def _render(self, fn, cid, review):
        p = C.c_void_p()
        self.engine._check(fn(self.engine.handle, self.handle, cid, review, C.byref(p)))
        s = C.string_at(p).decode()
        self.engine.lib.gk_free(p)
        return json.loads(s)

    def render(self, cid, review):
        return self._render(self.engine.lib.gk_render, cid, review)

    def render_error(self, cid, review):
        return self._render(self.engine.lib.gk_render_error, cid, review)

    def topk(self, k):
        """Per bitmap row of the most recent eval(): the k violating reviews with the smallest object key
        (group, version, kind, namespace, name), ties on the k-th key included -> {constraint id: (reviews, overflow)}."""
        out = C.POINTER(L.gk_topk_out)()
        self.engine._check(self.engine.lib.gk_table_topk(self.engine.handle, self.handle, k, C.byref(out)))
        o = out.contents
        res = {}
        for i in range(o.n_constraints):
            n = o.counts[i]
            res[int(o.constraint_ids[i])] = ([int(o.reviews[i * o.stride + j]) for j in range(n)], bool(o.overflow[i]))
        self.engine.lib.gk_topk_free(out)
        return res

    def stats(self):
        """host-side cost of building the table (gk_table_get_stats) as a dict"""
        st = L.gk_table_stats()
        self.engine._check(self.engine.lib.gk_table_get_stats(self.handle, C.byref(st)))
        return {f: getattr(st, f) for f, _ in st._fields_ if f != "reserved"}

    def totals(self):
        """Result-level totals of the most recent eval(): {constraint id: (results, violating pairs)} (gk_table_totals)."""
        out = C.POINTER(L.gk_totals_out)()
        self.engine._check(self.engine.lib.gk_table_totals(self.engine.handle, self.handle, C.byref(out)))
        o = out.contents
        res = {int(o.constraint_ids[i]): (int(o.results[i]), int(o.pairs[i])) for i in range(o.n_constraints)}
        self.rendered_pairs = int(o.rendered_pairs)   # violating pairs the host had to render (the device counted the others)
        self.engine.lib.gk_totals_free(out)
        return res

    def free(self):
        if self.handle:
            self.engine.lib.gk_table_free(self.handle)
            self.handle = None

    def __del__(self):
        try:
            self.free()
        except Exception:
            pass


class Engine:
    """OO view of gk_engine."""

    def __init__(self, device=0, elem_cap=None, hostemu=None, referential=True, gather_stats=False, tracing=False, disabled_builtins=None):
        """referential / gather_stats / tracing / disabled_builtins: the rego.Arg surface (gk_opts.flags, .disabled_builtins);
        disabled_builtins=None keeps the deployment's default {"http.send"}"""
        self.lib = L.load(hostemu)
        self.hostemu = bool(hostemu)
        opts = L.gk_opts()
        opts.device = device
        if elem_cap:
            for i, v in enumerate(elem_cap):
                opts.elem_cap[i] = v
        opts.flags = (0 if referential else L.GK_OPT_NO_REFERENTIAL) | (L.GK_OPT_GATHER_STATS if gather_stats else 0) | (L.GK_OPT_TRACE if tracing else 0)
        if disabled_builtins is not None:
            names = [n.encode() for n in disabled_builtins]
            self._disabled = (C.c_char_p * max(len(names), 1))(*names)
            opts.disabled_builtins, opts.n_disabled_builtins = self._disabled, len(names)
        h = C.c_void_p()
        self._check(self.lib.gk_engine_create(C.byref(opts), C.byref(h)))
        self.handle = h

    def _check(self, rc):
        if rc != L.GK_OK:
            msg = self.lib.gk_last_error().decode()
            if rc == L.GK_ERR_UNSUPPORTED:
                raise UnsupportedError(rc, msg)
            raise EngineError(rc, msg)

    def close(self):
        if getattr(self, "handle", None):
            self.lib.gk_engine_destroy(self.handle)
            self.handle = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def add_template(self, kind, rego, libs=()):
        arr = (C.c_char_p * max(1, len(libs)))(*[l.encode() for l in libs])
        self._check(self.lib.gk_template_add(self.handle, kind.encode(), rego.encode(), arr, len(libs)))

    def remove_template(self, kind):
        self._check(self.lib.gk_template_remove(self.handle, kind.encode()))

    def add_constraint(self, constraint):
        body = json.dumps(constraint).encode()
        cid = C.c_uint32()
        self._check(self.lib.gk_constraint_add(self.handle, body, len(body), C.byref(cid)))
        return cid.value

    def remove_constraint(self, kind, name):
        self._check(self.lib.gk_constraint_remove(self.handle, kind.encode(), name.encode()))

    def put_data(self, path, obj):
        arr = (C.c_char_p * len(path))(*[p.encode() for p in path])
        body = json.dumps(obj).encode()
        self._check(self.lib.gk_data_put(self.handle, arr, len(path), body, len(body)))

    def remove_data(self, path):
        arr = (C.c_char_p * len(path))(*[p.encode() for p in path])
        self._check(self.lib.gk_data_remove(self.handle, arr, len(path)))

    @staticmethod
    def _process_flag(process):
        if process is None:
            return 0
        if process not in ("audit", "webhook"):
            raise ValueError("tables are built for process 'audit' or 'webhook', not %r" % (process,))
        return L.GK_TABLE_PROCESS_AUDIT if process == "audit" else L.GK_TABLE_PROCESS_WEBHOOK

    def set_excluder(self, match_entries):
        """process.Excluder.Replace(New().Add(entries)): entries = the Config resource's spec.match"""
        js = json.dumps(match_entries or []).encode()
        self._check(self.lib.gk_excluder_replace(self.handle, js, len(js)))

    def is_excluded(self, process, review):
        """process.Excluder.IsNamespaceExcluded for one review (a to_review_in value)"""
        arr = (L.gk_review_in * 1)()
        self._fill(arr[0], review)
        out = C.c_int32(0)
        self._check(self.lib.gk_excluder_excluded(self.handle, process.encode(), arr, C.byref(out)))
        return bool(out.value)

    @staticmethod
    def _fill(a, r):
        a.kind, a.source = r.kind, r.source
        a.json, a.json_len = r.json, len(r.json)
        if r.namespace is not None:
            a.namespace_json, a.namespace_len = r.namespace, len(r.namespace)
        if r.ns_object is not None:
            a.ns_object_json, a.ns_object_len = r.ns_object, len(r.ns_object)
        a.operation = r.operation

    def create_table(self, reviews, keep_docs=True, resident=False, process=None, pre_matched=False):
        """process: 'audit' / 'webhook' applies the process excluder -- excluded reviews get status GK_REVIEW_EXCLUDED;
        pre_matched: GK_TABLE_PRE_MATCHED (the caller ran Matcher.Match: violation sets alone, no autoreject)"""
        n = len(reviews)
        arr = (L.gk_review_in * max(1, n))()
        keep = []
        for i, r in enumerate(reviews):
            a = arr[i]
            a.kind, a.source = r.kind, r.source
            a.json, a.json_len = r.json, len(r.json)
            keep.append(r)
            if r.namespace is not None:
                a.namespace_json, a.namespace_len = r.namespace, len(r.namespace)
            if r.ns_object is not None:
                a.ns_object_json, a.ns_object_len = r.ns_object, len(r.ns_object)
            a.operation = r.operation
        st = (C.c_int32 * max(1, n))()
        h = C.c_void_p()
        flags = (L.GK_TABLE_KEEP_DOCS if keep_docs else 0) | (L.GK_TABLE_RESIDENT if resident else 0) | self._process_flag(process) | (
            L.GK_TABLE_PRE_MATCHED if pre_matched else 0)
        self._check(self.lib.gk_table_create(self.handle, arr, n, flags, st, C.byref(h)))
        return Table(self, h, list(st)[:n], n)

    def create_table_spool(self, api_cache_dir, kind, folders, keep_docs=False, resident=True, process=None, keep_text=False):
        """gk_table_create_spool: ONE table of the objects pkg/audit spooled for `kind` under <api_cache_dir>/<kind>_<i>/
        (manager.go:519-551), reviewed the way reviewObjects does (manager.go:667-776).  -> (Table, info dict); info["names"][i]
        is the spool file of review i."""
        info, h = C.POINTER(L.gk_spool_info)(), C.c_void_p()
        flags = (L.GK_TABLE_KEEP_DOCS if keep_docs else 0) | (L.GK_TABLE_RESIDENT if resident else 0) | self._process_flag(process) | (L.GK_TABLE_KEEP_TEXT if keep_text else 0)
        self._check(self.lib.gk_table_create_spool(self.handle, str(api_cache_dir).encode(), kind.encode(), int(folders), flags, C.byref(info), C.byref(h)))
        o = info.contents
        n = int(o.n_reviews)
        d = {"n_files": int(o.n_files), "n_reviews": n, "n_unreadable": int(o.n_unreadable), "n_namespace_missing": int(o.n_namespace_missing),
             "n_folders_missing": int(o.n_folders_missing),
             "bytes": int(o.bytes), "names": [o.names[i].decode() for i in range(n)], "n_rejected": int(o.n_rejected), "n_excluded": int(o.n_excluded)}
        statuses = [int(o.statuses[i]) for i in range(n)]   # HandleReview's verdict per spooled object: the fail-closed loops read them
        self.lib.gk_spool_info_free(info)
        return Table(self, h, statuses, n), d

    def create_table_native(self, reviews_ptr, n, keep_docs=False, resident=False, process=None, keep_text=False, pruned=False):
        """gk_table_create on an existing gk_review_in array (e.g. synth.NativeBatch.reviews).  keep_text: the table remembers
        where the JSON text lives (the batch must outlive it) and parses only the reviews it has to render (totals, messages)"""
        st = (C.c_int32 * max(1, n))()
        h = C.c_void_p()
        flags = ((L.GK_TABLE_KEEP_DOCS if keep_docs else 0) | (L.GK_TABLE_RESIDENT if resident else 0) | self._process_flag(process) | (L.GK_TABLE_KEEP_TEXT if keep_text else 0)
                 | (L.GK_TABLE_PRUNED if pruned else 0))   # pruned: rows of the key paths the loaded constraints read, nothing else (include/gkgpu.h)
        self._check(self.lib.gk_table_create(self.handle, reviews_ptr, n, flags, st, C.byref(h)))
        return Table(self, h, st, n)

    def dump(self):
        p = C.c_void_p()
        self._check(self.lib.gk_dump(self.handle, C.byref(p)))
        s = C.string_at(p).decode()
        self.lib.gk_free(p)
        return s


# ---------------------------------------------------------------------------------------------- types.Result etc.
class Result:
    """frameworks types.Result"""

    def __init__(self, msg, constraint, details=None, enforcement_action="deny", scoped_actions=None, target=TARGET_NAME):
        self.target = target
        self.msg = msg
        self.constraint = constraint
        self.metadata = {"details": details if details is not None else {}}
        self.enforcement_action = enforcement_action
        self.scoped_enforcement_actions = scoped_actions

    def key(self):
        c = self.constraint
        return (c.get("kind"), (c.get("metadata") or {}).get("name"), self.msg, json.dumps(self.metadata, sort_keys=True),
                self.enforcement_action, tuple(self.scoped_enforcement_actions or ()))

    def __repr__(self):
        return "Result(%r, %s/%s, %s)" % (self.msg, self.constraint.get("kind"),
                                          (self.constraint.get("metadata") or {}).get("name"), self.enforcement_action)


class QueryResponse:
    """drivers.QueryResponse{Results, Trace *string, StatsEntries} (pkg/drivers/k8scel/driver.go:162-251)"""

    def __init__(self, results, stats_entries=None, trace=None):
        self.results = results
        self.stats_entries = stats_entries or []
        self.trace = trace


class ClientError(Exception):
    pass


def template_source(ct):
    """-> (kind, target name, rego, libs) of a ConstraintTemplate with exactly one target.  The Rego of a target comes
    from its `code` entry with engine "Rego" when there is one, else from the legacy `rego` / `libs` fields
    (website/docs/constrainttemplates.md:216-232)."""
    spec = ct.get("spec")
    if not isinstance(spec, dict):
        raise ClientError("invalid ConstraintTemplate: spec must be an object")
    names = spec
    for step in ("crd", "spec", "names"):
        names = names.get(step) if isinstance(names, dict) else None
    if not isinstance(names, dict) or "kind" not in names:
        raise ClientError("invalid ConstraintTemplate: missing spec.crd.spec.names.kind")
    targets = spec.get("targets") or []
    if len(targets) != 1:
        raise ClientError("invalid ConstraintTemplate: expected exactly 1 target, got %d" % len(targets))
    target = targets[0]
    rego_entries = [e.get("source") or {} for e in (target.get("code") or []) if e.get("engine") == "Rego"]
    source = rego_entries[-1] if rego_entries and rego_entries[-1].get("rego") is not None else target
    return names["kind"], target.get("target"), source.get("rego"), list(source.get("libs") or [])


class Driver:
    """drivers.Driver backed by the MI355X engine."""

    RUN_TIME_NS = "templateRunTimeNS"
    CONSTRAINT_COUNT = "constraintCount"

    def __init__(self, device=0, gather_stats=False, hostemu=None, elem_cap=None, tracing=False, referential=True, disabled_builtins=None, print_enabled=False):
        """the rego.Arg surface of the reference's construction sites: rego.Tracing(b), rego.GatherStats(), rego.Externs("inventory")
        (referential=True: --enable-referential-rules, main.go:479-484), rego.DisableBuiltins(names...) (main.go:424;
        None = the deployment's default {"http.send"}), rego.PrintEnabled (a label of the stats entries only)"""
        self.engine = Engine(device, elem_cap=elem_cap, hostemu=hostemu, referential=referential, gather_stats=gather_stats, tracing=tracing,
                             disabled_builtins=disabled_builtins)
        self.gather_stats = gather_stats
        self.tracing = tracing
        self.print_enabled = print_enabled
        self._ids = {}   # (kind, name) -> engine constraint id

    def Name(self):
        return "Rego"

    def AddTemplate(self, ct):
        kind, _, rego, libs = template_source(ct)
        if not rego:
            raise ClientError("template %s has no Rego source" % kind)
        self.engine.add_template(kind, rego, libs)

    def RemoveTemplate(self, ct):
        kind, _, _, _ = template_source(ct)
        try:
            self.engine.remove_template(kind)
        except EngineError as e:
            if e.code != L.GK_ERR_NOT_FOUND:
                raise
        for k in [k for k in self._ids if k[0].lower() == kind.lower()]:
            del self._ids[k]

    def AddConstraint(self, constraint):
        cid = self.engine.add_constraint(constraint)
        self._ids[(constraint.get("kind", ""), (constraint.get("metadata") or {}).get("name", ""))] = cid
        return cid

    def RemoveConstraint(self, constraint):
        key = (constraint.get("kind", ""), (constraint.get("metadata") or {}).get("name", ""))
        if key in self._ids:
            self.engine.remove_constraint(*key)
            del self._ids[key]

    def AddData(self, target, path, data):
        self.engine.put_data(list(path), data)

    def RemoveData(self, target, path):
        self.engine.remove_data(list(path))

    def constraint_id(self, constraint):
        key = (constraint.get("kind", ""), (constraint.get("metadata") or {}).get("name", ""))
        if key not in self._ids:
            raise EngineError(L.GK_ERR_NOT_FOUND, "unknown constraint template validator: %s" % constraint.get("kind"))
        return self._ids[key]

    def Query(self, target, constraints, review, namespace=None, stats_enabled=False, tracing=False):
        """drivers.Driver.Query as the reference defines it (pkg/drivers/k8scel/driver.go:162-251; frameworks drivers/rego): the
        caller -- Client.Review -- has ALREADY run Matcher.Match (pkg/target/matcher.go:21-42); `constraints` are evaluated and never
        matched again, no autoreject comes out of here.  What goes down is what a Go driver can reach: the AdmissionRequest of the
        review (ARGetter) and the reviews.Namespace option -- NOT gkReview.namespace / .source (unexported, pkg/target/review.go:16-21).
        gk_query_ex2(constraint ids, GK_QUERY_PRE_MATCHED), through the engine's micro-batcher: safe to call from many threads at
        once -- the webhook's concurrency, pkg/webhook/policy.go:142-146 -- and concurrent calls share one table and one launch."""
        return self._query(target, constraints, review, namespace, stats_enabled, tracing, True)

    def QueryMatching(self, target, constraints, review, namespace=None, stats_enabled=False, tracing=False):
        """The engine's own admission entry for callers that own namespace + source (this Python mirror, batch / audit code): Match AND
        violation are evaluated on the device for the listed constraints, a failed Matcher.Match comes back as an autoreject row
        (Result.msg "unable to match constraints: ...") -- Client.Review's three steps in one launch."""
        return self._query(target, constraints, review, namespace, stats_enabled, tracing, False)

    def _query(self, target, constraints, review, namespace, stats_enabled, tracing, pre_matched):
        rin = to_review_in(review, namespace)
        if rin is None:
            raise EngineError(L.GK_ERR_INVALID, "cannot convert review to ARGetter")
        arr = (L.gk_review_in * 1)()
        a = arr[0]
        a.kind, a.json, a.json_len, a.operation = rin.kind, rin.json, len(rin.json), rin.operation
        if not pre_matched:   # (a pre-matched Query hands over neither: a Go driver cannot read them)
            a.source = rin.source
            if rin.namespace is not None:
                a.namespace_json, a.namespace_len = rin.namespace, len(rin.namespace)
        if rin.ns_object is not None:
            a.ns_object_json, a.ns_object_len = rin.ns_object, len(rin.ns_object)
        wanted = {self.constraint_id(c): c for c in constraints}
        ids = (C.c_uint32 * max(1, len(wanted)))(*sorted(wanted))
        out, trace_p, st = C.c_void_p(), C.c_void_p(), L.gk_query_stats()
        rc = self.engine.lib.gk_query_ex2(self.engine.handle, arr, ids, len(wanted), (L.GK_QUERY_TRACE if tracing else 0) | (L.GK_QUERY_PRE_MATCHED if pre_matched else 0),
                                          C.byref(out), C.byref(trace_p), C.byref(st))
        if rc == L.GK_ERR_LIMIT:
            raise LimitError()
        if rc == L.GK_ERR_REVIEW:
            raise EngineError(rc, "review rejected by HandleReview: " + self.engine.lib.gk_last_error().decode())
        self.engine._check(rc)
        rows = json.loads(C.string_at(out).decode())
        self.engine.lib.gk_free(out)
        results = [Result(v["msg"], wanted[v["constraint"]], v.get("details", {})) for v in rows if v["constraint"] in wanted]
        self.last_query_stats = {"batch_size": st.batch_size, "queue_us": st.queue_us, "device_us": st.device_us, "total_us": st.total_us}
        trace = None
        if trace_p.value:
            trace = C.string_at(trace_p).decode()
            self.engine.lib.gk_free(trace_p)
        # instrumentation.StatsEntry per template kind, in the Rego driver's shape (pkg/gator/test/test_test.go:357-391): scope "template",
        # statsFor the kind, stats templateRunTimeNS + constraintCount with source {engine, Rego}, labels TracingEnabled / PrintEnabled /
        # target.  One launch answers every kind of the batch: its device time is what each entry reports (never 0: the reference's
        # test requires a run time).
        stats = []
        if self.gather_stats or stats_enabled:
            src = {"type": "engine", "value": self.Name()}
            run_ns = max(1, int(st.device_us * 1e3) or int(st.total_us * 1e3))
            by_kind = {}
            for c in constraints:
                by_kind[c.get("kind", "")] = by_kind.get(c.get("kind", ""), 0) + 1
            for kind in sorted(by_kind):
                stats.append({"scope": "template", "statsFor": kind,
                              "stats": [{"name": self.RUN_TIME_NS, "value": run_ns, "source": src},
                                        {"name": self.CONSTRAINT_COUNT, "value": by_kind[kind], "source": src}],
                              "labels": [{"name": "TracingEnabled", "value": bool(self.tracing or tracing)}, {"name": "PrintEnabled", "value": bool(self.print_enabled)},
                                         {"name": "target", "value": target}]})
        return QueryResponse(results, stats, trace)

    def ResidentSweep(self, result_totals=False):
        """gk_resident_sweep: bring the HBM-resident set (everything AddData'd) up to date and evaluate all constraints over
        it.  -> dict(n_objects, n_chunks, flattened, beyond_limits, sync_s, eval_s, pairs {constraint id: n}, results {...})"""
        out = C.POINTER(L.gk_sweep_out)()
        self.engine._check(self.engine.lib.gk_resident_sweep(self.engine.handle, L.GK_SWEEP_RESULT_TOTALS if result_totals else 0, C.byref(out)))
        o = out.contents
        res = {"n_objects": int(o.n_objects), "n_chunks": int(o.n_chunks), "flattened": int(o.flattened), "beyond_limits": int(o.beyond_limits),
               "sync_s": o.sync_s, "eval_s": o.eval_s,
               "pairs": {int(o.constraint_ids[i]): int(o.pairs[i]) for i in range(o.n_constraints)},
               "results": ({int(o.constraint_ids[i]): int(o.results[i]) for i in range(o.n_constraints)} if o.results else None)}
        self.engine.lib.gk_sweep_free(out)
        return res

    def ResidentReviewPreMatched(self, path, constraints):
        """gk_resident_review_ex(GK_QUERY_PRE_MATCHED): the violation sets of `constraints` (the caller matched them) for one resident
        object -- its resident text evaluated again through the admission batcher.  None when the object is not resident."""
        arr = (C.c_char_p * len(path))(*[p.encode() for p in path])
        wanted = sorted(self.constraint_id(c) for c in constraints)
        ids = (C.c_uint32 * max(1, len(wanted)))(*wanted)
        out = C.c_void_p()
        rc = self.engine.lib.gk_resident_review_ex(self.engine.handle, arr, len(path), ids, len(wanted), L.GK_QUERY_PRE_MATCHED, C.byref(out))
        if rc == L.GK_ERR_NOT_FOUND:
            return None
        if rc == L.GK_ERR_LIMIT:
            raise LimitError("/".join(path))
        self.engine._check(rc)
        rows = json.loads(C.string_at(out).decode())
        self.engine.lib.gk_free(out)
        return rows

    def ResidentReview(self, path):
        """gk_resident_review: the swept answer of one resident object (rows of gk_query's JSON), or None when the object is
        not in the swept set (unknown / changed since the sweep / policies changed)."""
        arr = (C.c_char_p * len(path))(*[p.encode() for p in path])
        out = C.c_void_p()
        rc = self.engine.lib.gk_resident_review(self.engine.handle, arr, len(path), C.byref(out))
        if rc == L.GK_ERR_NOT_FOUND:
            return None
        if rc == L.GK_ERR_LIMIT:
            raise LimitError("/".join(path))
        self.engine._check(rc)
        rows = json.loads(C.string_at(out).decode())
        self.engine.lib.gk_free(out)
        return rows

    def StartBatcher(self, max_batch=64, window_us=200, workers=0):
        """gk_batcher_start: how many concurrent Query calls share a launch, how long the first one waits for company, and
        how many batches may be in progress at once (0 = the engine's default, 2)"""
        opts = L.gk_batch_opts(max_batch, window_us, workers, 0)
        self.engine._check(self.engine.lib.gk_batcher_start(self.engine.handle, C.byref(opts)))

    def Dump(self):
        return self.engine.dump()

    def GetDescriptionForStat(self, stat_name):
        """drivers.Driver.GetDescriptionForStat (pkg/drivers/k8scel/driver.go:257-264): the stats of QueryResponse.stats_entries"""
        desc = {self.RUN_TIME_NS: "the number of nanoseconds it took to evaluate all constraints for a template (here: the device time of the launch the review's batch shared)",
                self.CONSTRAINT_COUNT: "the number of constraints that were evaluated for the given constraint kind"}
        if stat_name in desc:
            return desc[stat_name]
        raise ClientError("unknown stat name for Rego: %s" % stat_name)


# ---------------------------------------------------------------------------------------------- Client mirror
_KNOWN_ACTIONS = frozenset(("deny", "dryrun", "warn", "scoped"))


def get_enforcement_action(c):
    """util.GetEnforcementAction (pkg/util/enforcement_action.go:132-151; table: enforcement_action_test.go:113-165).
    No spec or an empty action -> deny; a known action -> itself; any other string -> "unrecognized"; a spec or an
    action of the wrong JSON type -> error."""
    if "spec" not in c or c["spec"] is None:
        return "deny"
    spec = c["spec"]
    action = spec.get("enforcementAction", "") if isinstance(spec, dict) else None
    if not isinstance(action, str):
        raise ClientError("unable to parse spec.enforcementAction")
    if not action:
        return "deny"
    return action if action in _KNOWN_ACTIONS else "unrecognized"


def scoped_actions_for_ep(ep, c):
    """util.ScopedActionForEP (pkg/util/enforcement_action.go:153-174; table: enforcement_action_test.go:235-385): one
    entry per scopedEnforcementActions element that lists `ep` or "*" among its enforcementPoints."""
    spec = c.get("spec")
    entries = spec.get("scopedEnforcementActions") if isinstance(spec, dict) else None
    if entries is None:
        return []
    if not isinstance(entries, list) or any(not isinstance(e, dict) for e in entries):
        raise ClientError("could not convert JSON to scopedEnforcementActions")
    wanted = (ep, ALL_EP)
    return [e.get("action") for e in entries
            if any(isinstance(pt, dict) and pt.get("name") in wanted for pt in (e.get("enforcementPoints") or []))]


def _fill_defaults(schema, node):
    """structural-schema defaulting of one node (k8s apiextensions `default`): object properties, additionalProperties
    and array items, depth first"""
    if not isinstance(schema, dict):
        return node
    if isinstance(node, list):
        item_schema = schema.get("items")
        return [_fill_defaults(item_schema, x) for x in node] if isinstance(item_schema, dict) else node
    if not isinstance(node, dict):
        return node
    declared = schema.get("properties") or {}
    for name, sub in declared.items():
        if name not in node and isinstance(sub, dict) and "default" in sub:
            node[name] = copy.deepcopy(sub["default"])
    extra = schema.get("additionalProperties")
    for name in list(node):
        if name in declared:
            node[name] = _fill_defaults(declared[name], node[name])
        elif isinstance(extra, dict):
            node[name] = _fill_defaults(extra, node[name])
    return node


def apply_schema_defaults(ct, c):
    """Client.AddConstraint applies the template CRD's OpenAPI-v3 defaults before the driver sees the constraint
    (pinned by test/gator/test/test.bats:277-291 through tests/test_oracle_rego.py)."""
    schema = ct
    for step in ("spec", "crd", "spec", "validation", "openAPIV3Schema"):
        schema = schema.get(step) if isinstance(schema, dict) else None
    if not isinstance(schema, dict):
        return c
    out = copy.deepcopy(c)
    if not isinstance(out.get("spec"), dict):
        if "default" not in json.dumps(schema):
            return out
        out["spec"] = {}
    spec = out["spec"]
    if "parameters" not in spec:
        if "default" not in schema:
            return out
        spec["parameters"] = copy.deepcopy(schema["default"])
    spec["parameters"] = _fill_defaults(schema, spec["parameters"])
    return out


def process_data(obj):
    """K8sValidationTarget.ProcessData (pkg/target/target.go:40-79) -> inventory path"""
    api = obj.get("apiVersion", "") if isinstance(obj.get("apiVersion"), str) else ""
    kind = obj.get("kind", "") if isinstance(obj.get("kind"), str) else ""
    md = obj.get("metadata") if isinstance(obj.get("metadata"), dict) else {}
    name = md.get("name", "") if isinstance(md.get("name"), str) else ""
    ns = md.get("namespace", "") if isinstance(md.get("namespace"), str) else ""
    version = api.split("/")[-1] if api.count("/") <= 1 else ""
    if version == "":
        raise ClientError("invalid request object: resource %s has no version" % name)
    if kind == "":
        raise ClientError("invalid request object: resource %s has no kind" % name)
    gv = api if "/" in api and not api.startswith("/") else version
    if ns == "":
        return ["cluster", gv, kind, name]
    return ["namespace", ns, gv, kind, name]


class ReviewFailure(ClientError):
    """One review of a batch that could not be evaluated: HandleReview rejected it (ErrReview) or it is beyond the
    engine's limits (`.cause` is then a LimitError).  ReviewBatch returns it IN PLACE of that review's result list, the
    way the reference's audit loop logs a Review error and moves on to the next object (pkg/audit/manager.go:717-726)."""

    def __init__(self, index, msg, cause=None):
        super().__init__(msg)
        self.index = index
        self.cause = cause


class AuditReport(dict):
    """AuditAggregate's answer: {(kind, apiVersion, name): {"total", "total_pairs", "violations"}} plus
    .totals_per_action ({enforcement action: results}, manager.go:904) and .errors ([ReviewFailure])."""

    def __init__(self):
        super().__init__()
        self.totals_per_action = {}
        self.errors = []

    def constraint_status(self, key, timestamp, violations_limit=20):
        """What updateConstraintStatus writes under the constraint's `status` (pkg/audit/manager.go:980-1034): auditTimestamp,
        totalViolations and -- only when there are any -- violations: the kept entries popped from the max-heap, i.e. in
        DESCENDING SVQueue order, each in StatusViolation's JSON shape (manager.go:100-109; namespace and
        enforcementActions are omitempty)."""
        ent = self.get(key) or {"total": 0, "violations": []}
        out = []
        for v in reversed(ent["violations"]):
            if len(out) >= violations_limit:
                break
            e = {"group": v["group"], "version": v["version"], "kind": v["kind"], "name": v["name"]}
            if v.get("namespace"):
                e["namespace"] = v["namespace"]
            e["message"] = v["message"]
            e["enforcementAction"] = v["enforcementAction"]
            if v.get("enforcementActions"):
                e["enforcementActions"] = list(v["enforcementActions"])
            out.append(e)
        st = {"auditTimestamp": timestamp, "totalViolations": int(ent["total"])}
        if out:
            st["violations"] = out
        return st


class Client:
    """constraintclient.Client for the single K8sValidationTarget, evaluating on the device."""

    def __init__(self, driver=None, enforcement_points=(WEBHOOK_EP, AUDIT_EP, GATOR_EP), hostemu=None):
        self.driver = driver or Driver(hostemu=hostemu)
        self.templates = {}
        self.constraints = {}   # (kind, name) -> constraint (defaulted)
        self.cached = {}        # inventory path -> object (everything AddData'd: the resident set)
        self.enforcement_points = tuple(enforcement_points)

    def AddTemplate(self, ct):
        kind, target, _, _ = template_source(ct)
        name = (ct.get("metadata") or {}).get("name", "")
        if name != kind.lower():
            raise ClientError("the ConstraintTemplate's name must be the lowercase of kind: got %r for kind %r" % (name, kind))
        if target != TARGET_NAME:
            raise ClientError("unknown target %r" % target)
        try:
            self.driver.AddTemplate(ct)
        except EngineError as e:
            if isinstance(e, UnsupportedError):
                raise
            raise ClientError(str(e))
        self.templates[kind.lower()] = ct

    def RemoveTemplate(self, ct):
        kind, _, _, _ = template_source(ct)
        self.driver.RemoveTemplate(ct)
        self.templates.pop(kind.lower(), None)
        for k in [k for k in self.constraints if k[0].lower() == kind.lower()]:
            del self.constraints[k]

    def AddConstraint(self, c, validate=True):
        kind = c.get("kind", "")
        if kind.lower() not in self.templates:
            raise ClientError("missing ConstraintTemplate: %s" % kind)   # ErrMissingConstraintTemplate
        if validate:
            validate_constraint(c)   # the target handler's check (target.go:185-219); validate=False: Match-layer error-path tests
        check_matcher(c)             # ToMatcher (target.go:246-261): always
        c = apply_schema_defaults(self.templates[kind.lower()], c)
        self.driver.AddConstraint(c)
        self.constraints[(kind, (c.get("metadata") or {}).get("name", ""))] = c

    def RemoveConstraint(self, c):
        self.driver.RemoveConstraint(c)
        self.constraints.pop((c.get("kind", ""), (c.get("metadata") or {}).get("name", "")), None)

    def AddData(self, obj):
        path = process_data(obj)
        # nsCache.Add (pkg/target/ns_cache.go:23-44): a core/v1 Namespace must convert into the typed object
        if path[:3] == ["cluster", "v1", "Namespace"]:
            for f in ("metadata", "spec", "status"):
                if obj.get(f) is not None and not isinstance(obj[f], dict):
                    raise ClientError("cannot cache type: cannot cache Namespace: %s must be an object" % f)
            md = obj.get("metadata") or {}
            for f in ("labels", "annotations"):
                v = md.get(f)
                if v is not None and not (isinstance(v, dict) and all(isinstance(x, str) for x in v.values())):
                    raise ClientError("cannot cache type: cannot cache Namespace: metadata.%s must be a map of strings" % f)
        self.driver.AddData(TARGET_NAME, path, dict(obj))
        self.cached[tuple(path)] = obj

    def RemoveData(self, obj):
        path = process_data(obj)
        self.driver.RemoveData(TARGET_NAME, path)
        self.cached.pop(tuple(path), None)

    def AuditFromCache(self):
        """pkg/audit's auditFromCache (manager.go:591-642) against the resident set: ONE sweep over everything that was
        AddData'd, then every cached object's results are read from the sweep's bitmap column (rendered only where a bit
        is set) -- the reference issues one serial Review per object.  -> ({path tuple: [Result]}, sweep dict)"""
        sweep = self.driver.ResidentSweep()
        info = self._active(AUDIT_EP)
        out = {}
        for path in self.cached:
            try:
                rows = self.driver.ResidentReview(list(path))
            except LimitError as e:
                out[path] = ReviewFailure(-1, str(e), e)
                continue
            if rows is None:
                raise EngineError(L.GK_ERR_INTERNAL, "cached object %r missing from the swept resident set" % (path,))
            res = []
            for v in rows:
                if v["constraint"] in info:
                    c, ea, scoped = info[v["constraint"]]
                    res.append(Result(v["msg"], c, {} if v.get("autoreject") else v.get("details", {}), ea, scoped))
            out[path] = res
        return out, sweep

    def _active(self, enforcement_point):
        """{engine constraint id: (constraint, enforcement action, scoped actions)} of the constraints enforced at this
        enforcement point (a `scoped` constraint without an action for the point is skipped, enforcement_action.go:153-174)"""
        info = {}
        for c in self.constraints.values():
            ea = get_enforcement_action(c)
            scoped = None
            if ea == "scoped":
                scoped = scoped_actions_for_ep(enforcement_point, c)
                if not scoped:
                    continue
            info[self.driver.constraint_id(c)] = (c, ea, scoped)
        return info

    def SetExcluder(self, match_entries):
        """The Config resource's spec.match -> the process excluder (pkg/controller/config/process/excluder.go:52-82);
        honoured by AuditAggregate / AuditFromCache (process "audit") and ReviewBatch(process=...)."""
        self.driver.engine.set_excluder(match_entries)

    def IsNamespaceExcluded(self, process, obj, namespace=None):
        """process.Excluder.IsNamespaceExcluded (excluder.go:96-105); AdmissionRequests the way the webhook looks at them"""
        r = to_review_in(obj, namespace)
        return False if r is None else self.driver.engine.is_excluded(process, r)

    def ReviewBatch(self, objs, enforcement_point=AUDIT_EP, namespaces=None, process=None):
        """Review many objects in ONE device launch (the shape the audit sweep takes).  -> one entry per object:
        list[Result], or a ReviewFailure for an object that HandleReview rejects or that is beyond the engine's limits
        (never an empty list for those: the engine fails closed).  process ('audit' / 'webhook'): objects the Config
        excludes for that process are skipped before evaluation ([]), as the audit manager / webhook handler do."""
        rins, idx = [], []
        for i, o in enumerate(objs):
            r = to_review_in(o, namespaces[i] if namespaces else None)
            if r is not None:
                rins.append(r)
                idx.append(i)
        out = [[] for _ in objs]
        if not rins:
            return out
        table = self.driver.engine.create_table(rins, process=process)
        try:
            ev = table.eval()
            failed = set()
            for k, st in enumerate(table.statuses):
                if st not in (L.GK_OK, L.GK_REVIEW_EXCLUDED):
                    failed.add(k)
                    out[idx[k]] = ReviewFailure(idx[k], "review %d rejected by HandleReview" % idx[k])   # ErrReview
            for k in ev.too_big_reviews():
                failed.add(k)
                out[idx[k]] = ReviewFailure(idx[k], str(LimitError("review %d" % idx[k])), LimitError("review %d" % idx[k]))
            info = self._active(enforcement_point)
            for cid, r in ev.pairs("err"):
                if cid in info and r not in failed:
                    c, ea, scoped = info[cid]
                    for v in table.render_error(cid, r):
                        out[idx[r]].append(Result(v["msg"], c, {}, ea, scoped))
            for cid, r in ev.pairs("viol"):
                if cid in info and r not in failed:
                    c, ea, scoped = info[cid]
                    rendered = table.render(cid, r)
                    if not rendered:
                        raise EngineError(L.GK_ERR_INTERNAL, "device flagged (constraint %d, review %d) but the template renders "
                                                             "no violation: device / renderer disagree" % (cid, r))
                    for v in rendered:
                        out[idx[r]].append(Result(v["msg"], c, v.get("details", {}), ea, scoped))
            return out
        finally:
            table.free()

    def Review(self, obj, enforcement_point=AUDIT_EP, namespace=None):
        res = self.ReviewBatch([obj], enforcement_point, [namespace])[0]
        if isinstance(res, ReviewFailure):
            raise res
        return res

    def AuditAggregate(self, objs, namespaces=None, limit=20, msg_size=256):
        """pkg/audit/manager.go:885-941 (addAuditResponsesToUpdateLists) for one resident set: ONE device sweep; per
        constraint the RESULT total (totalViolationsPerConstraint, :902: one per types.Result -- gk_table_totals), the
        totals per enforcement action (:904) and the `limit` smallest violations (LimitQueue order: group, version, kind,
        namespace, name, message, action; messages truncated to msg_size bytes, :1039-1048).  Only the top-k candidates
        selected on the device are rendered to messages for the lists.  -> AuditReport"""
        rins = [to_review_in(o, namespaces[i] if namespaces else None) for i, o in enumerate(objs)]
        table = self.driver.engine.create_table(rins, resident=True, process="audit")
        report = AuditReport()
        try:
            ev = table.eval()
            for k, st in enumerate(table.statuses):
                if st not in (L.GK_OK, L.GK_REVIEW_EXCLUDED):
                    report.errors.append(ReviewFailure(k, "review %d rejected by HandleReview" % k))
            for k in ev.too_big_reviews():
                report.errors.append(ReviewFailure(k, str(LimitError("review %d" % k)), LimitError("review %d" % k)))
            top = table.topk(limit)
            totals = table.totals()
            row = {int(cid): i for i, cid in enumerate(ev.constraint_ids)}
            for cid, (c, ea, scoped) in self._active(AUDIT_EP).items():
                if cid not in row:
                    continue
                reviews, overflow = top.get(cid, ([], False))
                if overflow:   # more key ties than the device list holds: walk the bitmap row (exact, rare)
                    reviews = [int(r) for r in EvalResult.bits(ev.viol[row[cid]], ev.n_reviews)]
                cand = []
                for r in reviews:
                    obj = review_object(objs[r])
                    g, ver, k = obj_gvk(obj)
                    for v in table.render(cid, r):
                        cand.append({"group": g, "version": ver, "kind": k, "namespace": (obj.get("metadata") or {}).get("namespace", "") or "",
                                     "name": (obj.get("metadata") or {}).get("name", "") or "", "message": truncate_string(v["msg"], msg_size),
                                     "enforcementAction": ea, "enforcementActions": scoped})
                cand.sort(key=lambda x: tuple(s_.encode("utf-8") for s_ in (x["group"], x["version"], x["kind"], x["namespace"], x["name"],
                                                                           x["message"], x["enforcementAction"])))
                key = (c.get("kind", ""), c.get("apiVersion", ""), (c.get("metadata") or {}).get("name", ""))
                n_results, n_pairs = totals.get(cid, (0, 0))
                report[key] = {"total": n_results, "total_pairs": n_pairs, "violations": cand[:limit]}
                if n_results:
                    report.totals_per_action[ea] = report.totals_per_action.get(ea, 0) + n_results
            return report
        finally:
            table.free()
