"""debug: which reviews take the large-capacity kernel variant on the GPU (test_edge_cases saw 2 instead of 1)"""
import os, sys, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from gatekeeper_amd import driver as D, synth
fx = synth.load_fixtures()
for mode in ("jit", "interp"):
    if mode == "interp":
        os.environ["GK_NO_JIT"] = "1"
    c = D.Client(D.Driver(hostemu=(len(sys.argv) > 1 and sys.argv[1] == "hostemu")))
    for t in synth.psp_templates(fx):
        c.AddTemplate(t)
    for k in synth.psp_constraints():
        c.AddConstraint(k)
    objs = synth.gen_objects(65, seed=11)
    big = {"apiVersion": "v1", "kind": "Pod", "metadata": {"name": "big", "namespace": "prod-01"}, "spec": {
        "containers": [{"name": "c%d" % i, "image": "x", "securityContext": {"privileged": i == 37},
                        "volumeMounts": [{"name": "v%d" % i, "mountPath": "/m", "readOnly": i % 2 == 0}]} for i in range(40)],
        "volumes": [{"name": "v%d" % i, "hostPath": {"path": "/foo/x%d" % i}} for i in range(40)]}}
    sets = {"o0": [objs[0]], "big": [big], "o1": [objs[1]], "o0+big": [objs[0], big], "big+o1": [big, objs[1]], "o0+o1": [objs[0], objs[1]],
            "all": [objs[0], big, objs[1]], "o1+big+o0": [objs[1], big, objs[0]]}
    for name, os_ in sets.items():
        rins = [D.to_review_in(D.AugmentedUnstructured(D.Unstructured(o), None, "Original")) for o in os_]
        t = c.driver.engine.create_table(rins)
        ev = t.eval()
        print(mode, name, "n_overflow", ev.n_overflow, "too_big", ev.too_big_reviews(), "viol", int(ev.counts.sum()))
        t.free()
