"""Ad-hoc campaign of tests/test_template_fuzz.py::test_random_constraint_sets_in_one_plan over a range of seeds: a dozen random templates +
constraints loaded TOGETHER (shared sub-formulas, dictionary predicates of different templates on the same leaves, element carriers one
template registers and another one meets), product vs oracle.  usage: python tools/scratch/multi_template_fuzz.py FIRST LAST [backend]"""
import os, sys, random
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
for p in (ROOT, os.path.join(ROOT, "tests"), os.path.join(ROOT, "tests", "golden")): sys.path.insert(0, p)
os.environ.setdefault("GK_RENDER_CHECK", "1")
import test_template_fuzz as F
from gatekeeper_amd import driver as D
from oracle import client as OC, target as OT
from parity_util import make_client
first, last = int(sys.argv[1]), int(sys.argv[2])
backend = sys.argv[3] if len(sys.argv) > 3 else "hostemu-gen"
bad = tot = loaded = 0
for seed in range(first, last + 1):
    F.ENVELOPE = seed % 2 == 1
    rng = random.Random(seed)
    objs = [F.rand_obj(rng, i) for i in range(14)]
    for g in range(3):
        c, oc = make_client(backend), OC.Client()
        for i in range(12):
            rego, kind = F.template(rng, g * 100 + i), "K8sFuzz%dx%d" % (g, i)
            params = {"p": rng.choice(["x", 1, True]), "q": rng.choice(["yy", 2]), "allowed": rng.sample(["x", "yy", 1, 2, True], 2)}
            match = rng.choice([None, {"kinds": [{"apiGroups": [""], "kinds": ["Pod"]}]}, {"namespaces": ["d"]}, {"excludedNamespaces": ["d"]}, {"name": "o1*"}])
            k = {"apiVersion": "constraints.gatekeeper.sh/v1beta1", "kind": kind, "metadata": {"name": "c"}, "spec": {"parameters": params}}
            if match: k["spec"]["match"] = match
            try:
                c.AddTemplate(F.tmpl(kind, rego)); c.AddConstraint(k)
            except D.UnsupportedError:
                c.RemoveTemplate(F.tmpl(kind, rego)); continue
            oc.add_template(F.tmpl(kind, rego)); oc.add_constraint(k); loaded += 1
        try:
            got = c.ReviewBatch(F.mk_reviews(D, objs, seed), D.GATOR_EP)
        except Exception as e:
            bad += 1; print("seed", seed, "group", g, "ERROR", str(e)[:200]); sys.stdout.flush(); continue
        for j, rv in enumerate(F.mk_reviews(OT, objs, seed)):
            if isinstance(got[j], Exception): continue
            want = sorted((r.constraint["kind"], r.msg) for r in oc.review(rv, OC.GATOR_EP))
            g_ = sorted((r.constraint["kind"], r.msg) for r in got[j])
            tot += len(want)
            if g_ != want:
                bad += 1; print("seed", seed, "group", g, "review", j, "DIFF", g_, want); sys.stdout.flush()
print("seeds %d..%d backend %s: %d constraints loaded, %d results compared, %d differences" % (first, last, backend, loaded, tot, bad))
