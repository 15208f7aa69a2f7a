"""Ad-hoc campaign of tests/test_parity.py test_structural_fuzz over a range of seeds: mutated objects through flattener, compiled
predicates and renderer against the oracle.  usage: [GK_FORCE_PRUNE=1] [GK_NO_INDEX=1] python tools/scratch/structural_fuzz.py FIRST LAST [psp|corpus]"""
import os, sys, json
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
for p in (ROOT, os.path.join(ROOT, "tests"), os.path.join(ROOT, "tests", "golden")): sys.path.insert(0, p)
os.environ.setdefault("GK_RENDER_CHECK", "1")
import test_parity as TP
from parity_util import load_both, assert_parity
from gatekeeper_amd import driver as D, synth
fx = json.load(open(os.path.join(ROOT, "tests", "golden", "reference_fixtures.json")))
nss = synth.gen_namespaces()
first, last = int(sys.argv[1]), int(sys.argv[2])
policy = sys.argv[3] if len(sys.argv) > 3 else "psp"
if policy == "psp":
    ts, cs = synth.psp_templates(fx), synth.audit_constraints()
else:
    ts, cs = synth.corpus(fx); ts = ts[::4]; kinds = {t["spec"]["crd"]["spec"]["names"]["kind"] for t in ts}; cs = [c for c in cs if c["kind"] in kinds]
tot = 0; bad = 0
for seed in range(first, last + 1):
    c, oc = load_both("hostemu", ts, cs)
    rng = synth.SplitMix64(seed)
    revs = []
    for o in synth.gen_objects(160, seed=seed, mixed=True):
        m = TP._mutate(rng, TP._mutate(rng, o))
        if not isinstance(m, dict): m = {"apiVersion": "v1", "kind": "Pod", "metadata": {"name": "m"}}
        if not (isinstance(m.get("kind"), str) and m["kind"]): m["kind"] = "Pod"
        md = m.get("metadata")
        ns = synth.namespace_for(m, nss) if isinstance(md, dict) and isinstance(md.get("namespace"), str) else None
        revs.append(D.AugmentedUnstructured(D.Unstructured(m), ns, "Original"))
    try:
        refused = []
        tot += assert_parity(c, oc, revs, refused=refused)
    except AssertionError as e:
        bad += 1; print("seed", seed, "FAILED", str(e)[:300]); sys.stdout.flush()
print("seeds %d..%d policy %s prune=%s index=%s: %d results compared, %d failing seeds" % (first, last, policy, os.environ.get("GK_FORCE_PRUNE"), not os.environ.get("GK_NO_INDEX"), tot, bad))
