"""split() / trim() on review strings at every length class of the device's byte-position masks (vm_core.hpp SplitMask): inline
strings (<= 7 bytes), header-only heap strings (<= 12), one / two / three 16-byte blocks beyond the header, the last word
(61..64 bytes) and the byte-wise path beyond 64 bytes; separators and trimmed bytes first, last, doubled, absent, at the block
borders 11|12, 27|28, 43|44, 59|60, 63|64.  Product vs oracle: rendered results and raw bitmaps, interpreter and generated
plan source, emulator and MI355X."""
import random

import pytest

from gatekeeper_amd import driver as D
from gatekeeper_amd import synth
from parity_util import BACKENDS, assert_parity, load_both


def tmpl(kind, rego):
    return {"apiVersion": "templates.gatekeeper.sh/v1", "kind": "ConstraintTemplate", "metadata": {"name": kind.lower()},
            "spec": {"crd": {"spec": {"names": {"kind": kind}}}, "targets": [{"target": "admission.k8s.gatekeeper.sh", "rego": rego}]}}


T = {
    # last / first / second component, component count (the library's image-tag and registry idioms)
    "K8sSplitTag": '''package k
violation[{"msg": msg}] {
  c := input.review.object.spec.containers[_]
  parts := split(c.image, ":")
  parts[count(parts) - 1] == "latest"
  msg := sprintf("latest tag on %v", [c.name])
}
violation[{"msg": msg}] {
  c := input.review.object.spec.containers[_]
  count(split(c.image, ":")) == 1
  msg := sprintf("no tag on %v", [c.name])
}
violation[{"msg": msg}] {
  c := input.review.object.spec.containers[_]
  split(c.image, "/")[0] == "docker.io"
  msg := sprintf("docker hub image on %v", [c.name])
}
violation[{"msg": msg}] {
  c := input.review.object.spec.containers[_]
  split(c.image, "/")[1] == "library"
  msg := sprintf("library image on %v", [c.name])
}
violation[{"msg": msg}] {
  c := input.review.object.spec.containers[_]
  count(split(c.image, "/")) > 3
  msg := sprintf("deep repository on %v", [c.name])
}
''',
    # trim + split: path components (hostPath prefixes)
    "K8sSplitPath": '''package k
violation[{"msg": msg}] {
  v := input.review.object.spec.volumes[_]
  p := split(trim(v.hostPath.path, "/"), "/")
  p[0] == "var"
  p[1] == "lib"
  msg := sprintf("var/lib hostPath %v", [v.name])
}
violation[{"msg": msg}] {
  v := input.review.object.spec.volumes[_]
  p := split(trim(v.hostPath.path, "/"), "/")
  p[count(p) - 1] == "sock"
  msg := sprintf("socket hostPath %v", [v.name])
}
violation[{"msg": msg}] {
  v := input.review.object.spec.volumes[_]
  count(split(trim(v.hostPath.path, "/"), "/")) >= 5
  msg := sprintf("deep hostPath %v", [v.name])
}
''',
}

BORDERS = (0, 1, 6, 7, 8, 11, 12, 13, 15, 16, 27, 28, 29, 43, 44, 45, 59, 60, 61, 63, 64, 65, 66, 80, 130)


def strings(rng, sep, words):
    """strings of every border length with separators in interesting places"""
    out = []
    for n in BORDERS:
        base = "".join(rng.choice("abcdefghij") for _ in range(n))
        out.append(base)
        for at in {0, n - 1, n // 2, 11, 12, 27, 28, 43, 44, 59, 60, 63, 64}:
            if 0 <= at < n:
                out.append(base[:at] + sep + base[at + 1:])
        if n >= 2:
            out.append(sep + base[1:-1] + sep)
            out.append(sep * n)
            k = rng.randrange(n)
            out.append(base[:k] + sep + sep + base[k + 2:] if k + 2 <= n else base)
        for w in words:   # the constants the templates compare with, as first / last / middle component, padded to length n
            pad = "x" * max(0, n - len(w) - 1)
            out += [w + sep + pad, pad + sep + w, (pad[:len(pad) // 2] + sep + w + sep + pad[len(pad) // 2:])]
    return out


def pods(strs, field):
    objs = []
    for i in range(0, len(strs), 5):
        chunk = strs[i:i + 5]
        if field == "image":
            spec = {"containers": [{"name": "c%d" % j, "image": s} for j, s in enumerate(chunk)]}
        else:
            spec = {"containers": [{"name": "c", "image": "x"}], "volumes": [{"name": "v%d" % j, "hostPath": {"path": s}} for j, s in enumerate(chunk)]}
        objs.append({"apiVersion": "v1", "kind": "Pod", "metadata": {"name": "p%d" % (i // 5), "namespace": "default"}, "spec": spec})
    return objs


@pytest.mark.parametrize("backend", BACKENDS)
def test_split_and_trim_at_every_length_class(backend):
    rng = random.Random(7)
    c, oc = load_both(backend, [tmpl(k, r) for k, r in T.items()],
                      [{"apiVersion": "constraints.gatekeeper.sh/v1beta1", "kind": k, "metadata": {"name": "x"}, "spec": {}} for k in T])
    imgs = strings(rng, ":", ["latest", "docker.io"]) + strings(rng, "/", ["docker.io", "library"])
    imgs += ["docker.io/library/nginx:latest", "docker.io/library/" + "n" * 40 + ":latest", "registry.example.com:5000/team/app/sub/img:1.2.3", ":", "/", "a:b:c:latest",
             "docker.io/" + "r" * 53 + ":latest", "x" * 58 + ":latest", "x" * 57 + ":latest", "x" * 100 + ":latest", "docker.io/library/" + "z" * 100]
    paths = strings(rng, "/", ["var", "lib", "sock"])
    paths += ["/var/lib/kubelet", "var/lib", "///var/lib///", "/var/lib/" + "d" * 60, "/run/containerd/containerd.sock/", "/a/b/c/d/sock", "/" * 70, "/var/" + "q" * 55 + "/sock",
              "/" + "/".join("d%d" % i for i in range(30)), "/var/lib/" + "/".join("e" * 7 for _ in range(6)) + "/sock"]
    objs = pods(imgs, "image") + pods(paths, "path")
    n = assert_parity(c, oc, [D.AugmentedUnstructured(D.Unstructured(o), None, "Original") for o in objs])
    assert n > 200


@pytest.mark.parametrize("backend", BACKENDS)
def test_path_prefixes_at_every_length_class(backend, fixtures):
    """K8sPSPHostFilesystem's path_matches (trim + split + component-wise prefix: the fused P_SPLIT_PREFIX predicate) on hostPath
    strings of every length class, prefixes that end at the block borders, trailing / doubled slashes"""
    rng = random.Random(11)
    t_ = [t for t in synth.psp_templates(fixtures) if t["spec"]["crd"]["spec"]["names"]["kind"] == "K8sPSPHostFilesystem"]
    deep = "/srv/" + "k" * 22          # a prefix whose last byte is string byte 26: the separator test reads position 27 | 28
    cons = [{"apiVersion": "constraints.gatekeeper.sh/v1beta1", "kind": "K8sPSPHostFilesystem", "metadata": {"name": "hp%d" % i},
             "spec": {"parameters": {"allowedHostPaths": a}}} for i, a in enumerate((
                 [{"pathPrefix": "/var/lib"}], [{"pathPrefix": "/var/lib/"}, {"pathPrefix": "/run", "readOnly": True}], [{"pathPrefix": deep}],
                 [{"pathPrefix": "/" + "p" * 58}], [{"pathPrefix": "/" + "p" * 62}], [{"pathPrefix": "/" + "p" * 70}], [{"pathPrefix": "/"}]))]
    c, oc = load_both(backend, t_, cons)
    paths = strings(rng, "/", ["var", "lib", "run"])
    for stem in ("/var/lib", "/run", deep, "/" + "p" * 58, "/" + "p" * 62, "/" + "p" * 70):
        paths += [stem, stem + "/", stem + "x", stem + "/x", stem + "//x", "/" + stem, stem[:-1], stem + "/" + "y" * 40, stem + "/" + "y" * 3, "//" + stem.strip("/") + "//"]
    objs = []
    for i in range(0, len(paths), 4):
        vols = [{"name": "v%d" % j, "hostPath": {"path": s}} for j, s in enumerate(paths[i:i + 4])]
        objs.append({"apiVersion": "v1", "kind": "Pod", "metadata": {"name": "p%d" % i, "namespace": "default"},
                     "spec": {"containers": [{"name": "c", "image": "x", "volumeMounts": [{"name": v["name"], "mountPath": "/m", "readOnly": j % 2 == 0} for j, v in enumerate(vols)]}],
                              "volumes": vols}})
    n = assert_parity(c, oc, [D.AugmentedUnstructured(D.Unstructured(o), None, "Original") for o in objs])
    assert n > 100
