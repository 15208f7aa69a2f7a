"""Synthetic workloads for BASELINE.json's configs: deterministic SplitMix64 generator.

  config[1]  30 PSP constraints (5 in-tree PSP templates x 6 parameterisations) x N synthetic Pod reviews
  config[2]  50 constraints (the 30 + 20 with heavier match blocks) x N mixed cluster objects (audit sweep)

The templates themselves are the reference's own policy fixtures (pkg/webhook/testdata/psp-all-violations/psp-templates/*.yaml,
demo/agilebank/templates/*.yaml), shipped as data in gatekeeper_amd/data/policy_templates.json because /root/reference does
not exist on the GPU box.  Constraint parameterisations and objects are generated here.
Shared by bench.py, __graft_entry__.smoke() and tests/ so that every leg sees identical inputs.
"""
from __future__ import annotations

import json
import os

SEED = 0x6B8E9C4A
_ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
PSP_DIR = "pkg/webhook/testdata/psp-all-violations/"


class SplitMix64:
    def __init__(self, seed=SEED):
        self.s = seed & 0xFFFFFFFFFFFFFFFF

    def next(self):
        self.s = (self.s + 0x9E3779B97F4A7C15) & 0xFFFFFFFFFFFFFFFF
        z = self.s
        z = ((z ^ (z >> 30)) * 0xBF58476D1CE4E5B9) & 0xFFFFFFFFFFFFFFFF
        z = ((z ^ (z >> 27)) * 0x94D049BB133111EB) & 0xFFFFFFFFFFFFFFFF
        return z ^ (z >> 31)

    def uniform(self):
        return (self.next() >> 11) / float(1 << 53)

    def below(self, n):
        return self.next() % n

    def chance(self, p):
        return self.uniform() < p

    def pick(self, seq):
        return seq[self.below(len(seq))]

    def weighted(self, pairs):
        u = self.uniform()
        acc = 0.0
        for v, p in pairs:
            acc += p
            if u < acc:
                return v
        return pairs[-1][0]


def load_fixtures():
    """the reference's policy templates the synthetic workloads use (policy INPUTS shipped as data with the package:
    gatekeeper_amd/data/policy_templates.json, written by tests/golden/make_golden.py from /root/reference)"""
    with open(os.path.join(_ROOT, "gatekeeper_amd", "data", "policy_templates.json"), encoding="utf-8") as fh:
        return json.load(fh)


# ------------------------------------------------------------------------------------------------ namespaces
NAMESPACES = (["kube-%s" % s for s in ("system", "public", "node-lease", "proxy", "dns")] +
              ["prod-%02d" % i for i in range(30)] + ["dev-%02d" % i for i in range(30)] +
              ["team-%02d" % i for i in range(35)])
LABEL_KEYS = ["app", "tier", "env", "team", "owner", "release", "track", "zone", "region", "cost-center", "app.kubernetes.io/name",
              "app.kubernetes.io/part-of", "app.kubernetes.io/managed-by", "chart", "heritage", "component", "role", "version",
              "stage", "project", "squad", "domain", "criticality", "pci", "gdpr", "tenant", "cluster", "shard", "canary",
              "backup", "monitored", "sidecar"]
LABEL_VALUES = ["a", "b", "prod", "dev", "web", "db", "cache", "true", "false", "blue", "green", "v1", "v2", "core", "edge", "x"]
HOST_PATHS = ["/tmp", "/foo", "/foo/bar", "/var/log", "/etc"]
IMAGES = ["nginx", "nginx:1.25", "openpolicyagent/opa:0.9.2", "gcr.io/proj/app:latest", "quay.io/org/tool:v3", "busybox"]


def gen_namespaces():
    """100 Namespace objects with deterministic labels (env / team / pci)."""
    rng = SplitMix64(SEED ^ 0x5A5A)
    out = {}
    for name in NAMESPACES:
        labels = {"kubernetes.io/metadata.name": name}
        labels["env"] = "prod" if name.startswith("prod") else "dev" if name.startswith("dev") else "shared"
        if rng.chance(0.5):
            labels["team"] = "team-%d" % rng.below(8)
        if rng.chance(0.2):
            labels["pci"] = "true"
        out[name] = {"apiVersion": "v1", "kind": "Namespace", "metadata": {"name": name, "labels": labels}}
    return out


def _labels(rng):
    n = 2 + rng.below(5)
    out = {}
    for _ in range(n):
        value = rng.pick(LABEL_VALUES)
        out[rng.pick(LABEL_KEYS)] = value
    return out


def _container(rng, idx, vol_names, init=False):
    c = {"name": ("init-%d" if init else "c%d") % idx, "image": rng.pick(IMAGES)}
    if rng.chance(0.6):
        sc = {}
        if rng.chance(0.05 / 0.6):
            sc["privileged"] = True
        elif rng.chance(0.3):
            sc["privileged"] = False
        if rng.chance(0.3):
            sc["runAsNonRoot"] = True
        if rng.chance(0.2):
            sc["allowPrivilegeEscalation"] = False
        c["securityContext"] = sc
    nports = rng.weighted([(0, 0.4), (1, 0.4), (2, 0.2)])
    if nports:
        ports = []
        for _ in range(nports):
            p = {"containerPort": 1 + rng.below(65535)}
            if rng.chance(0.05):
                p["hostPort"] = 1 + rng.below(65535)
            if rng.chance(0.3):
                p["protocol"] = "TCP"
            ports.append(p)
        c["ports"] = ports
    if vol_names:
        mounts = []
        for vn in vol_names:
            if rng.chance(0.6):
                m = {"name": vn, "mountPath": "/mnt/" + vn}
                if rng.chance(0.5):
                    m["readOnly"] = rng.chance(0.8)
                mounts.append(m)
        if mounts:
            c["volumeMounts"] = mounts
    if rng.chance(0.5):
        c["resources"] = {"limits": {"cpu": rng.pick(["100m", "200m", "1", "2"]), "memory": rng.pick(["128Mi", "1Gi", "2Gi"])}}
    if rng.chance(0.3):
        c["env"] = [{"name": "E%d" % k, "value": rng.pick(LABEL_VALUES)} for k in range(1 + rng.below(3))]
    return c


def _pod_spec(rng):
    nvol = rng.weighted([(0, 0.4), (1, 0.3), (2, 0.2), (3, 0.1)])
    vols = []
    for v in range(nvol):
        vt = rng.weighted([("configMap", 0.3), ("secret", 0.25), ("emptyDir", 0.25), ("hostPath", 0.1), ("persistentVolumeClaim", 0.1)])
        vol = {"name": "vol-%d" % v}
        if vt == "configMap":
            vol[vt] = {"name": "cm-%d" % rng.below(20)}
        elif vt == "secret":
            vol[vt] = {"secretName": "s-%d" % rng.below(20)}
        elif vt == "emptyDir":
            vol[vt] = {}
        elif vt == "hostPath":
            vol[vt] = {"path": rng.pick(HOST_PATHS)}
        else:
            vol[vt] = {"claimName": "pvc-%d" % rng.below(20)}
        vols.append(vol)
    names = [v["name"] for v in vols]
    nc = rng.weighted([(1, 0.5), (2, 0.3), (3, 0.15), (4, 0.05)])
    spec = {"containers": [_container(rng, i, names) for i in range(nc)]}
    if rng.chance(0.2):
        spec["initContainers"] = [_container(rng, 0, names, init=True)]
    if vols:
        spec["volumes"] = vols
    if rng.chance(0.03):
        spec["hostNetwork"] = True
    if rng.chance(0.02):
        spec["hostPID"] = True
    if rng.chance(0.02):
        spec["hostIPC"] = True
    if rng.chance(0.3):
        spec["serviceAccountName"] = "sa-%d" % rng.below(10)
    if rng.chance(0.3):
        spec["restartPolicy"] = "Always"
    return spec


def gen_pod(rng, i):
    ns = rng.pick(NAMESPACES)
    md = {"name": "pod-%07d" % i, "namespace": ns, "labels": _labels(rng)}
    if rng.chance(0.2):
        md["annotations"] = {"note": "generated"}
    return {"apiVersion": "v1", "kind": "Pod", "metadata": md, "spec": _pod_spec(rng)}


def object_rng(seed, i):
    """Every object has its own generator, seeded from (seed, index): any slice of the object stream can be produced
    independently (shards, samples, the native generator's threads) and is identical wherever it is produced."""
    return SplitMix64(SplitMix64((seed ^ (i * 0xD1342543DE82EF95)) & 0xFFFFFFFFFFFFFFFF).next())


def gen_object(seed, i, mixed=False):
    rng = object_rng(seed, i)
    kind = "Pod"
    if mixed:
        kind = rng.weighted([("Pod", 0.8), ("Deployment", 0.1), ("Namespace", 0.05), ("Service", 0.025), ("ConfigMap", 0.025)])
    if kind == "Pod":
        return gen_pod(rng, i)
    if kind == "Deployment":
        pod = gen_pod(rng, i)
        return {"apiVersion": "apps/v1", "kind": "Deployment",
                "metadata": {"name": "dep-%07d" % i, "namespace": pod["metadata"]["namespace"], "labels": pod["metadata"]["labels"]},
                "spec": {"replicas": 1 + rng.below(5), "template": {"metadata": {"labels": pod["metadata"]["labels"]}, "spec": pod["spec"]}}}
    if kind == "Namespace":
        return {"apiVersion": "v1", "kind": "Namespace", "metadata": {"name": "gen-ns-%07d" % i, "labels": _labels(rng)}}
    if kind == "Service":
        return {"apiVersion": "v1", "kind": "Service", "metadata": {"name": "svc-%07d" % i, "namespace": rng.pick(NAMESPACES)},
                "spec": {"ports": [{"port": 80 + rng.below(1000)}], "selector": {"app": rng.pick(LABEL_VALUES)}}}
    return {"apiVersion": "v1", "kind": "ConfigMap", "metadata": {"name": "cm-%07d" % i, "namespace": rng.pick(NAMESPACES)},
            "data": {"k%d" % k: rng.pick(LABEL_VALUES) for k in range(1 + rng.below(4))}}


def admission_request_for(obj, i):
    """the AdmissionRequest (CREATE) around Pod `i` of the stream, as the validating webhook receives it; the native
    generator (mixed = 2) writes the same document"""
    md = obj["metadata"]
    return {"uid": "uid-%d" % i, "kind": {"group": "", "version": "v1", "kind": "Pod"}, "resource": {"group": "", "version": "v1", "resource": "pods"},
            "name": md["name"], "namespace": md.get("namespace", ""), "operation": "CREATE",
            "userInfo": {"username": "system:serviceaccount:ci:deployer", "groups": ["system:serviceaccounts", "system:authenticated"]},
            "object": obj, "oldObject": None, "dryRun": False, "options": {"kind": "CreateOptions", "apiVersion": "meta.k8s.io/v1"}}


def gen_objects(n, seed=SEED, mixed=False, start=0):
    """objects [start, start + n) of the synthetic stream `seed`.  mixed=False: all Pods (config[1]).  mixed=True: 80% Pod,
    10% Deployment, 5% Namespace, 5% Service/ConfigMap (config[2], the audit sweep).  The native generator
    (csrc/synth.cpp, gk_synth_batch_create) produces the same objects as JSON text; tests/test_synth.py pins the two."""
    return [gen_object(seed, i, mixed) for i in range(start, start + n)]


# ------------------------------------------------------------------------------------------------ policies
_POD_KINDS = [{"apiGroups": [""], "kinds": ["Pod"]}]

_MATCH_VARIANTS = [
    {"kinds": _POD_KINDS},
    {"kinds": _POD_KINDS, "excludedNamespaces": ["kube-*"]},
    {"kinds": _POD_KINDS, "namespaces": ["prod-*", "team-0*"]},
    {"kinds": _POD_KINDS, "labelSelector": {"matchExpressions": [{"key": "canary", "operator": "DoesNotExist"}]}},
    {"kinds": _POD_KINDS, "namespaceSelector": {"matchLabels": {"env": "prod"}}},
    {"kinds": [{"apiGroups": ["*"], "kinds": ["*"]}], "scope": "Namespaced", "excludedNamespaces": ["*-system", "dev-1*"]},
]

_PARAMS = {
    "K8sPSPHostFilesystem": [
        {"allowedHostPaths": [{"readOnly": True, "pathPrefix": "/foo"}]},
        {"allowedHostPaths": [{"pathPrefix": "/var/log"}]},
        {"allowedHostPaths": []},
        {"allowedHostPaths": [{"pathPrefix": "/tmp"}, {"readOnly": True, "pathPrefix": "/etc"}]},
        {"allowedHostPaths": [{"readOnly": True, "pathPrefix": "/"}]},
        {"allowedHostPaths": [{"readOnly": False, "pathPrefix": "/foo/bar"}, {"pathPrefix": "/var"}]},
    ],
    "K8sPSPHostNamespace": [None] * 6,
    "K8sPSPHostNetworkingPorts": [
        {"hostNetwork": True, "min": 80, "max": 9000},
        {"hostNetwork": False, "min": 80, "max": 9000},
        {"hostNetwork": False, "min": 1, "max": 65535},
        {"hostNetwork": True, "min": 1024, "max": 32767},
        {"hostNetwork": False},
        {"hostNetwork": True, "min": 30000, "max": 30100},
    ],
    "K8sPSPPrivilegedContainer": [None] * 6,
    "K8sPSPVolumeTypes": [
        {"volumes": ["configMap", "emptyDir", "projected", "secret", "downwardAPI", "persistentVolumeClaim", "flexVolume"]},
        {"volumes": ["*"]},
        {"volumes": ["configMap", "secret"]},
        {"volumes": ["emptyDir", "hostPath", "persistentVolumeClaim"]},
        {"volumes": []},
        {"volumes": ["configMap", "emptyDir", "secret", "hostPath"]},
    ],
}


def psp_templates(fixtures=None):
    fx = fixtures or load_fixtures()
    return [fx["yaml"][p]["docs"][0] for p in sorted(fx["yaml"]) if p.startswith(PSP_DIR + "psp-templates/")]


def psp_constraints():
    """config[1]: 5 PSP template kinds x 6 parameterisations = 30 constraints."""
    out = []
    for kind in sorted(_PARAMS):
        for k, params in enumerate(_PARAMS[kind]):
            spec = {"match": _MATCH_VARIANTS[k]}
            if params is not None:
                spec["parameters"] = params
            out.append({"apiVersion": "constraints.gatekeeper.sh/v1beta1", "kind": kind,
                        "metadata": {"name": "%s-%d" % (kind.lower(), k)}, "spec": spec})
    return out


def audit_constraints():
    """config[2]: the 30 above + 20 with heavier match blocks (globs, selectors, multi-kind, enforcement actions)."""
    out = psp_constraints()
    kinds = sorted(_PARAMS)
    extra_match = [
        {"kinds": [{"apiGroups": [""], "kinds": ["Pod"]}, {"apiGroups": ["apps"], "kinds": ["Deployment"]}], "namespaces": ["*-0*"]},
        {"kinds": _POD_KINDS, "labelSelector": {"matchLabels": {"env": "prod"}}},
        {"kinds": _POD_KINDS, "labelSelector": {"matchExpressions": [{"key": "tier", "operator": "In", "values": ["web", "db"]}]}},
        {"kinds": _POD_KINDS, "labelSelector": {"matchExpressions": [{"key": "team", "operator": "NotIn", "values": ["a", "b"]},
                                                                     {"key": "app", "operator": "Exists"}]}},
        {"kinds": _POD_KINDS, "namespaceSelector": {"matchExpressions": [{"key": "pci", "operator": "Exists"}]}},
        {"kinds": _POD_KINDS, "namespaceSelector": {"matchLabels": {"env": "dev"}}, "excludedNamespaces": ["dev-2*"]},
        {"kinds": _POD_KINDS, "name": "pod-00*"},
        {"kinds": _POD_KINDS, "name": "*7"},
        {"kinds": _POD_KINDS, "scope": "Cluster"},
        {"kinds": _POD_KINDS, "namespaces": ["team-*"], "excludedNamespaces": ["team-1*", "team-2*"], "source": "Original"},
    ]
    for k in range(20):
        kind = kinds[k % 5]
        params = _PARAMS[kind][(k // 5) % 6]
        spec = {"match": extra_match[k % len(extra_match)]}
        if params is not None:
            spec["parameters"] = params
        if k % 4 == 1:
            spec["enforcementAction"] = "dryrun"
        elif k % 4 == 2:
            spec["enforcementAction"] = "warn"
        out.append({"apiVersion": "constraints.gatekeeper.sh/v1beta1", "kind": kind,
                    "metadata": {"name": "%s-x%d" % (kind.lower(), k)}, "spec": spec})
    return out


# ------------------------------------------------------------------------------------------------ config[4]: 200 templates
_FAMILIES = [   # (template file, [20 parameterisations])
    ("demo/agilebank/templates/k8srequiredlabels_template.yaml",
     [{"message": "label policy %d" % i, "labels": [{"key": k, "allowedRegex": rx}]} for i, (k, rx) in enumerate(
         [("owner", "^[a-z]+$"), ("owner", "^(a|b|x)$"), ("team", "^team-[0-9]+$"), ("env", "^(prod|dev)$"), ("app", "^[a-z0-9-]{1,8}$"), ("tier", "web|db"),
          ("owner", ".+\\.agilebank\\.demo$"), ("release", "^v[0-9]+$"), ("track", ""), ("zone", "^[a-c]$"), ("region", "^.{2,}$"), ("cost-center", "^[0-9]+$"),
          ("project", "^p"), ("squad", "d$"), ("domain", "(?i)^CORE$"), ("criticality", "\\bprod\\b"), ("pci", "^(true|false)$"), ("tenant", "^[^0-9]*$"),
          ("shard", "^x?$"), ("canary", "^.*$")])]),
    ("demo/agilebank/templates/k8sallowedrepos_template.yaml",
     [{"repos": r} for r in (["gcr.io/"], ["quay.io/", "gcr.io/"], ["openpolicyagent/"], ["nginx"], ["docker.io/library/"], ["gcr.io/proj/"], ["quay.io/org/"], ["busybox"],
                             ["k8s.gcr.io/", "registry.k8s.io/"], [""], ["gcr.io/proj/app"], ["quay.io/org/tool:v3"], ["n", "b"], ["ghcr.io/"], ["openpolicyagent/opa"],
                             ["gcr.io", "quay.io", "docker.io"], ["x"], ["nginx:1.2"], ["registry.internal:5000/"], ["quay.io/org/", "busybox", "nginx"])]),
    ("demo/agilebank/remediation/k8sbannedimagetags_template.yaml",
     [{"tags": t} for t in (["latest"], ["latest", "v3"], ["1.25"], ["0.9.2"], ["latest", "master", "main"], ["v3"], ["dev"], ["1.25", "latest"], ["stable"], ["edge", "canary"],
                            ["0.9.2", "v3", "1.25"], ["nightly"], ["rc"], ["latest", "1.25", "0.9.2", "v3"], ["beta"], ["alpha", "latest"], ["v1"], ["v2"], ["test"], ["snapshot", "latest"])]),
    ("demo/agilebank/templates/k8scontainterlimits_template.yaml",
     [{"cpu": c, "memory": m} for c, m in (("200m", "1Gi"), ("100m", "128Mi"), ("1", "1Gi"), ("2", "2Gi"), ("500m", "512Mi"), ("150m", "1G"), ("1000m", "1000Mi"), ("50m", "64Mi"),
                                           ("4", "8Gi"), ("250m", "256Mi"), ("300m", "1500M"), ("1", "1Ti"), ("199m", "1073741823"), ("2000m", "2048Mi"), ("3", "3Gi"),
                                           ("120m", "100Mi"), ("1", "999Mi"), ("800m", "1Gi"), ("101m", "129Mi"), ("10", "1Ei"))]),
    ("demo/agilebank/templates/k8srequiredprobes_template.yaml",
     [{"probes": pr, "probeTypes": pt} for pr, pt in ((["readinessProbe", "livenessProbe"], ["tcpSocket", "httpGet", "exec"]), (["readinessProbe"], ["httpGet"]),
                                                       (["livenessProbe"], ["tcpSocket", "httpGet", "exec"]), (["startupProbe"], ["exec"]))] * 5),
]
_CORPUS_MATCH = [
    {"kinds": _POD_KINDS, "namespaces": ["prod-*", "*-system"]},
    {"kinds": _POD_KINDS},
    {"kinds": _POD_KINDS, "excludedNamespaces": ["kube-*"]},
    {"kinds": [{"apiGroups": [""], "kinds": ["Pod"]}, {"apiGroups": ["apps"], "kinds": ["Deployment"]}], "namespaces": ["prod-*", "*-system"]},
    {"kinds": _POD_KINDS, "namespaceSelector": {"matchLabels": {"env": "prod"}}},
    {"kinds": _POD_KINDS, "labelSelector": {"matchExpressions": [{"key": "canary", "operator": "DoesNotExist"}]}, "namespaces": ["*-0*", "team-*"]},
]


def corpus(fixtures=None, n_templates=200):
    """BASELINE.json configs[4]: `n_templates` ConstraintTemplates + one constraint each -- the in-tree families (required
    labels with allowedRegex, allowed repos, banned image tags, container limits, required probes, the 5 PSP templates),
    every copy under its own kind, x parameterisations incl. regex allow-lists and `namespaces: ["prod-*", "*-system"]` globs.
    -> (templates, constraints)"""
    fx = fixtures or load_fixtures()
    fams = [(fx["yaml"][p]["docs"][0], params) for p, params in _FAMILIES]
    for kind in sorted(_PARAMS):
        t = next(t_ for t_ in psp_templates(fx) if t_["spec"]["crd"]["spec"]["names"]["kind"] == kind)
        fams.append((t, [_PARAMS[kind][i % 6] for i in range(20)]))
    templates, constraints = [], []
    i = 0
    while len(templates) < n_templates:
        t, params = fams[i % len(fams)]
        k = i // len(fams)
        base_kind = t["spec"]["crd"]["spec"]["names"]["kind"]
        kind = "%sV%02d" % (base_kind, k)
        ct = json.loads(json.dumps(t))
        ct["metadata"]["name"] = kind.lower()
        ct["spec"]["crd"]["spec"]["names"]["kind"] = kind
        templates.append(ct)
        spec = {"match": _CORPUS_MATCH[(i + k) % len(_CORPUS_MATCH)]}
        if params[k % len(params)] is not None:
            spec["parameters"] = params[k % len(params)]
        if i % 7 == 3:
            spec["enforcementAction"] = "dryrun"
        constraints.append({"apiVersion": "constraints.gatekeeper.sh/v1beta1", "kind": kind, "metadata": {"name": "%s-c" % kind.lower()}, "spec": spec})
        i += 1
    return templates, constraints


def namespace_for(obj, namespaces):
    """The *corev1.Namespace the audit loop attaches (pkg/audit/manager.go:694-705): looked up by the object's
    namespace; None for cluster-scoped objects or unknown namespaces."""
    ns = (obj.get("metadata") or {}).get("namespace")
    return namespaces.get(ns) if ns else None


class NativeBatch:
    """objects [start, start + n) of stream `seed` generated by the native generator (csrc/synth.cpp): JSON text plus the
    gk_review_in array, handed to Engine.create_table_native without touching Python objects."""

    def __init__(self, lib, n, seed=SEED, mixed=False, start=0, namespaces=None, requests=False, high_cardinality=False):
        """requests=True: the Pods wrapped in AdmissionRequest documents (review kind GK_REVIEW_ADMISSION_REQUEST);
        high_cardinality=True: every container's image tag and name unique in the stream (include/gksynth.h: ingest measurements)"""
        import ctypes as C
        self.lib = lib
        arr, k = None, 0
        if namespaces is not None:
            self._ns = [json.dumps(namespaces[name]).encode() for name in NAMESPACES]
            arr, k = (C.c_char_p * len(self._ns))(*self._ns), len(self._ns)
        h = C.c_void_p()
        rc = lib.gk_synth_batch_create(seed & 0xFFFFFFFFFFFFFFFF, start, n, (1 if mixed else 0) | (2 if requests else 0) | (16 if high_cardinality else 0), arr, k, C.byref(h))
        if rc != 0:
            raise RuntimeError("gk_synth_batch_create failed: %d" % rc)
        self.handle = h
        self.n = n

    @property
    def reviews(self):
        return self.lib.gk_synth_batch_reviews(self.handle)

    @property
    def json_bytes(self):
        return int(self.lib.gk_synth_batch_json_bytes(self.handle))

    def json_text(self, i):
        r = self.reviews[i]
        import ctypes as C
        return C.string_at(r.json, r.json_len)

    def namespace_text(self, i):
        r = self.reviews[i]
        import ctypes as C
        return C.string_at(r.namespace_json, r.namespace_len) if r.namespace_json else None

    def query_storm(self, engine, threads, per_thread, constraint_ids=None, pre_matched=False):
        """`threads` NATIVE threads x `per_thread` gk_query calls on this batch's reviews (include/gksynth.h) -> dict.
        constraint_ids: through gk_query_ex2 with that list (Driver.Query as the Go shim calls it), pre_matched: GK_QUERY_PRE_MATCHED"""
        import ctypes as C
        from . import _lib as L
        out = L.gk_storm_out()
        if constraint_ids is not None:
            ids = (C.c_uint32 * max(1, len(constraint_ids)))(*constraint_ids)
            rc = self.lib.gk_synth_query_storm_ex(engine.handle, self.handle, threads, per_thread, ids, len(constraint_ids), L.GK_QUERY_PRE_MATCHED if pre_matched else 0, out)
        else:
            rc = self.lib.gk_synth_query_storm(engine.handle, self.handle, threads, per_thread, out)
        if rc != 0:
            raise RuntimeError("gk_synth_query_storm failed: %d" % rc)
        d = {k: getattr(out, k) for k, _ in L.gk_storm_out._fields_}
        d["threads"] = threads
        d["reviews_per_s"] = d["calls"] / d["seconds"] if d["seconds"] > 0 else 0.0
        return d

    def free(self):
        if self.handle:
            self.lib.gk_synth_batch_free(self.handle)
            self.handle = None

    def __del__(self):
        try:
            self.free()
        except Exception:
            pass
