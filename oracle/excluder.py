"""TEST INFRASTRUCTURE (oracle): CPU restatement of the process excluder.

Only tests/ may import this.  Restates
  pkg/controller/config/process/excluder.go:52-76   Excluder.Add (the "*" process fans out to allProcesses, :31-36)
  pkg/controller/config/process/excluder.go:96-105  IsNamespaceExcluded (core Namespace objects match on their NAME)
  pkg/controller/config/process/excluder.go:120-128 exactOrWildcardMatch
  pkg/wildcard/wildcard.go:16-29                    Wildcard.Matches
  pkg/webhook/common.go:149-189                     skipExcludedNamespace (oldObject on DELETE, namespace := request.namespace)
  pkg/audit/manager.go:599-608, 971-978             the audit loop skips excluded objects before Review
Pinned against the rows of pkg/controller/config/process/excluder_test.go (tests/test_excluder.py).
"""

ALL_PROCESSES = ("audit", "webhook", "mutation-webhook", "sync")


def wildcard_matches(w, candidate):
    pre, suf = w.startswith("*"), w.endswith("*")
    if pre and suf:
        inner = w[1:] if pre else w
        if inner.endswith("*"):
            inner = inner[:-1]
        return inner in candidate
    if pre:
        return candidate.endswith(w[1:])
    if suf:
        return candidate.startswith(w[:-1])
    return w == candidate


class Excluder:
    def __init__(self, entries=()):
        self.excluded = {}
        self.add(entries)

    def add(self, entries):
        for ent in entries or ():
            for ns in ent.get("excludedNamespaces") or ():
                for op in ent.get("processes") or ():
                    for proc in (ALL_PROCESSES if op == "*" else (op,)):
                        self.excluded.setdefault(proc, set()).add(ns)

    def get_excluded_namespaces(self, process):
        return sorted(self.excluded.get(process, ()))

    def match(self, process, ns):
        return any(wildcard_matches(w, ns) for w in self.excluded.get(process, ()))

    def is_namespace_excluded(self, process, obj):
        """obj: an unstructured object (dict)"""
        av = obj.get("apiVersion") or ""
        group = av.split("/")[0] if av.count("/") == 1 else ""
        md = obj.get("metadata") if isinstance(obj.get("metadata"), dict) else {}
        if obj.get("kind") == "Namespace" and group == "":
            return self.match(process, md.get("name") or "")
        return self.match(process, md.get("namespace") or "")

    def webhook_skips(self, process, request):
        """skipExcludedNamespace for an AdmissionRequest dict; a decode error is reported as 'not excluded' (the handler
        logs it and reviews the request)"""
        data = request.get("oldObject") if request.get("operation") == "DELETE" else request.get("object")
        if not isinstance(data, dict) or not isinstance(data.get("kind"), str) or not data["kind"]:
            return False
        obj = dict(data)
        md = dict(obj.get("metadata") or {}) if isinstance(obj.get("metadata"), dict) else {}
        md["namespace"] = request.get("namespace") or ""
        obj["metadata"] = md
        return self.is_namespace_excluded(process, obj)
