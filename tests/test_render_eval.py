"""The host renderer's two evaluators (csrc/ceval.cpp: concrete, tried first; csrc/pe.cpp: the partial evaluator on a concrete
review, the fallback): the same violation sets -- tests/conftest.py switches the cross-check on for the whole suite -- and the
concrete one really is the one that serves (a silent fallback for everything would make the cross-check vacuous)."""
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

SCRIPT = r"""
import sys
from gatekeeper_amd import driver as D, synth
fx = synth.load_fixtures()
drv = D.Driver(device=0, hostemu=True); client = D.Client(drv)
t, c = synth.corpus(fx)
for x in t[::3]: client.AddTemplate(x)
kinds = {x["spec"]["crd"]["spec"]["names"]["kind"] for x in t[::3]}
for x in c:
    if x["kind"] in kinds: client.AddConstraint(x)
for x in synth.psp_templates(fx): client.AddTemplate(x)
for x in synth.audit_constraints(): client.AddConstraint(x)
n = 300
nss = synth.gen_namespaces()
b = synth.NativeBatch(drv.engine.lib, n, seed=synth.SEED, mixed=True, start=0, namespaces=nss)
tb = drv.engine.create_table_native(b.reviews, n, keep_docs=True, resident=True)
ev = tb.eval()
pairs = 0
for row, cid in enumerate(ev.constraint_ids):
    for r in D.EvalResult.bits(ev.viol[row], ev.n_reviews):
        assert tb.render(int(cid), int(r))
        pairs += 1
print("PAIRS", pairs)
"""


def test_the_concrete_evaluator_serves_and_agrees_with_the_partial_one():
    env = dict(os.environ, GK_RENDER_CHECK="1", GK_RENDER_STATS="1", PYTHONPATH=ROOT)
    p = subprocess.run([sys.executable, "-c", SCRIPT], env=env, capture_output=True, text=True, timeout=600)
    assert p.returncode == 0, p.stderr[-2000:]
    pairs = int(p.stdout.split("PAIRS")[1].split()[0])
    assert pairs > 1500
    stats = [l for l in p.stderr.splitlines() if "[gkgpu render]" in l]
    assert stats, "no statistics line: nothing was rendered through Template::render"
    last = stats[-1].split()
    fast, slow = int(last[last.index("evaluator") + 1]), int(last[len(last) - 1 - last[::-1].index("evaluator") + 1])
    assert fast >= 1024 and slow * 50 <= fast, stats[-1]
