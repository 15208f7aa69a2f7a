"""ORACLE (test infrastructure only -- never imported by the product path).

CPU restatement of the audit aggregation step that follows the hot path:
  pkg/audit/manager.go:112-203    StatusViolation, SVQueue (max-heap order), LimitQueue
  pkg/audit/manager.go:885-941    addAuditResponsesToUpdateLists
  pkg/audit/manager.go:1039-1048  truncateString
  pkg/audit/manager.go:62-68      msgSize = 256, --constraint-violations-limit default 20
Pinned by pkg/audit/manager_test.go:41-103 (Test_SVQueue, Test_LimitQueue) and :231-273 (Test_truncateString).
"""
from __future__ import annotations

MSG_SIZE = 256
DEFAULT_VIOLATIONS_LIMIT = 20


def truncate_string(s: str, size: int) -> str:
    """manager.go:1039-1048 -- Go slices BYTES, so operate on the UTF-8 encoding."""
    b = s.encode("utf-8")
    if len(b) > size:
        if size > 3:
            size -= 3
        return b[:size].decode("utf-8", errors="surrogateescape") + "..."
    return s


def sv_key(v):
    """SVQueue.Less order (manager.go:118-138): group, version, kind, namespace, name, message, enforcementAction."""
    return (v["group"], v["version"], v["kind"], v.get("namespace", ""), v["name"], v["message"],
            v["enforcementAction"])


class LimitQueue:
    """Keeps the `limit` SMALLEST violations by sv_key (the heap pops the largest; manager.go:178-183)."""

    def __init__(self, limit=DEFAULT_VIOLATIONS_LIMIT):
        self.limit = limit
        self.items = []

    def push(self, v):
        self.items.append(v)
        while len(self.items) > self.limit:
            self.items.remove(max(self.items, key=lambda x: tuple(s.encode("utf-8") for s in sv_key(x))))

    def sorted(self):
        return sorted(self.items, key=lambda x: tuple(s.encode("utf-8") for s in sv_key(x)))


def add_audit_responses(update_lists, totals_per_constraint, totals_per_action, results, limit=DEFAULT_VIOLATIONS_LIMIT):
    """results: iterable of (Result, obj dict). Mirrors addAuditResponsesToUpdateLists (manager.go:885-941)."""
    from .match import obj_gvk, obj_name, obj_namespace
    for r, obj in results:
        c = r.constraint
        key = (c.get("kind", ""), c.get("apiVersion", ""), (c.get("metadata") or {}).get("name", ""))
        q = update_lists.get(key)
        if q is None:
            q = update_lists[key] = LimitQueue(limit)
        totals_per_constraint[key] = totals_per_constraint.get(key, 0) + 1
        totals_per_action[r.enforcement_action] = totals_per_action.get(r.enforcement_action, 0) + 1
        g, v, k = obj_gvk(obj)
        msg = r.msg
        if len(msg.encode("utf-8")) > MSG_SIZE:
            msg = truncate_string(msg, MSG_SIZE)
        q.push({"group": g, "version": v, "kind": k, "namespace": obj_namespace(obj), "name": obj_name(obj),
                "message": msg, "enforcementAction": r.enforcement_action,
                "enforcementActions": r.scoped_enforcement_actions})


def status_violations(queue: LimitQueue, violations_limit=DEFAULT_VIOLATIONS_LIMIT):
    """updateConstraintStatus (manager.go:980-1003): the LimitQueue is POPPED into status.violations -- a max-heap, so the
    list is in DESCENDING SVQueue order -- and each entry is StatusViolation's JSON (manager.go:100-109: namespace and
    enforcementActions are omitempty)."""
    out = []
    for v in reversed(queue.sorted()):
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
    return out


def constraint_status(queue: LimitQueue, total, timestamp, violations_limit=DEFAULT_VIOLATIONS_LIMIT):
    """the fields updateConstraintStatus writes under the constraint's `status` (manager.go:1004-1034): violations is
    REMOVED when there are none"""
    st = {"auditTimestamp": timestamp, "totalViolations": int(total)}
    v = status_violations(queue, violations_limit)
    if v:
        st["violations"] = v
    return st
