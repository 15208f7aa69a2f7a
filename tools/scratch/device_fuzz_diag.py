#!/usr/bin/env python3
"""Which template and object of a failing device-fuzz seed: device_fuzz_diag.py SEED MODE [backend]   (MODE = SEED % 4 as the campaign sets it)"""
import json, os, random, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import test_template_fuzz as F
from gatekeeper_amd import driver as D
seed = int(sys.argv[1]); mode = seed % 4; backend = sys.argv[3] if len(sys.argv) > 3 else "gpu"
F.ENVELOPE, F.NUMERIC = mode == 1, mode >= 2
rng = random.Random(seed)
objs = [F.rand_obj(rng, i) for i in range(14)]
cases = []
for i in range(60):
    rego, mk = F.template(rng, i), F.tmpl
    if mode == 3: rego, mk = F.to_v1(rego), F.tmpl_v1
    kind = "K8sFuzz%d" % i
    params = {"p": rng.choice(["x", 1, True]), "q": rng.choice(["yy", 2]), "allowed": rng.sample(["x", "yy", 1, 2, True], 2), "rules": [{"k": rng.choice(F.KEYS), "v": rng.choice(["x", 1, "yy"])} for _ in range(rng.randint(0, 2))]}
    cases.append((kind, mk(kind, rego), {"apiVersion": "constraints.gatekeeper.sh/v1beta1", "kind": kind, "metadata": {"name": "c"}, "spec": {"parameters": params}}, rego))
for lo in range(0, 60, 10):
    c = F.make_client(backend); ids = {}
    for kind, t, k, rego in cases[lo:lo + 10]:
        try:
            c.AddTemplate(t); c.AddConstraint(k)
        except D.UnsupportedError:
            c.RemoveTemplate(t); continue
        ids[c.driver.constraint_id(k)] = (kind, rego, k)
    rv = F.mk_reviews(D, objs, seed)
    try:
        c.ReviewBatch(rv, D.GATOR_EP)
        print("plan %d: ok (%d templates)" % (lo // 10, len(ids)))
    except D.EngineError as e:
        import re
        m = re.search(r"constraint (\d+), review (\d+)", str(e)); cid, r = int(m.group(1)), int(m.group(2))
        kind, rego, k = ids[cid]
        print("plan %d: %s\n--- %s parameters %s\n%s\n--- object %d: %s" % (lo // 10, e, kind, json.dumps(k["spec"]["parameters"]), rego, r, json.dumps(objs[r])))
        # the same template ALONE in a plan
        c1 = F.make_client(backend)
        t = [x for x in cases if x[0] == kind][0]
        c1.AddTemplate(t[1]); c1.AddConstraint(t[2])
        try:
            got = c1.ReviewBatch(rv, D.GATOR_EP); print("alone in a plan: ok, review %d -> %r" % (r, [x.msg for x in got[r]] if not isinstance(got[r], Exception) else got[r]))
        except D.EngineError as e2: print("alone in a plan:", e2)
        # greedy reduction: drop templates of the plan while the disagreement stays
        def fails(sel):
            cc = F.make_client(backend)
            for kind2, t2, k2, rego2 in sel:
                try:
                    cc.AddTemplate(t2); cc.AddConstraint(k2)
                except D.UnsupportedError:
                    cc.RemoveTemplate(t2)
            try:
                cc.ReviewBatch(rv, D.GATOR_EP); return False
            except D.EngineError:
                return True
        sel = list(cases[lo:lo + 10])
        assert fails(sel), "the whole plan does not fail when rebuilt"
        k = 0
        while k < len(sel):
            if sel[k][0] != kind and fails(sel[:k] + sel[k + 1:]): sel.pop(k)
            else: k += 1
        # ... and the objects
        keep = list(range(len(objs)))
        print("minimal failing set: %s" % [x[0] for x in sel])
        for kind2, t2, k2, rego2 in sel:
            if kind2 != kind: print("--- %s parameters %s\n%s" % (kind2, json.dumps(k2["spec"]["parameters"]), rego2))
