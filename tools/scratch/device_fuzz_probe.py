#!/usr/bin/env python3
"""The failing pair of a device-fuzz seed, looked at through the raw evaluation: device_fuzz_probe.py SEED KIND_A KIND_B [backend]"""
import json, os, random, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import test_template_fuzz as F
from gatekeeper_amd import driver as D
seed = int(sys.argv[1]); want = sys.argv[2:4]; backend = sys.argv[4] if len(sys.argv) > 4 else "gpu"
mode = seed % 4
F.ENVELOPE, F.NUMERIC = mode == 1, mode >= 2
rng = random.Random(seed)
objs = [F.rand_obj(rng, i) for i in range(14)]
cases = {}
for i in range(60):
    rego, mk = F.template(rng, i), F.tmpl
    if mode == 3: rego, mk = F.to_v1(rego), F.tmpl_v1
    kind = "K8sFuzz%d" % i
    params = {"p": rng.choice(["x", 1, True]), "q": rng.choice(["yy", 2]), "allowed": rng.sample(["x", "yy", 1, 2, True], 2), "rules": [{"k": rng.choice(F.KEYS), "v": rng.choice(["x", 1, "yy"])} for _ in range(rng.randint(0, 2))]}
    cases[kind] = (mk(kind, rego), {"apiVersion": "constraints.gatekeeper.sh/v1beta1", "kind": kind, "metadata": {"name": "c"}, "spec": {"parameters": params}})
def raw(kinds):
    c = F.make_client(backend)
    for k in kinds: c.AddTemplate(cases[k][0]); c.AddConstraint(cases[k][1])
    rv = F.mk_reviews(D, objs, seed)
    rins = [D.to_review_in(o, None) for o in rv]
    table = c.driver.engine.create_table(rins)
    ev = table.eval()
    ids = {c.driver.constraint_id(cases[k][1]): k for k in kinds}
    print("%s: n_overflow %s too_big %s viol %s" % (kinds, ev.n_overflow, list(ev.too_big_reviews()), sorted((ids[cid], r) for cid, r in ev.pairs("viol"))))
raw(want)
if not os.environ.get("GK_PROBE_FIRST_ONLY"): raw(want[::-1]); raw(want[:1]); raw(want[1:])
