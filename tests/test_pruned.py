"""GK_TABLE_PRUNED (round 4): a table that serves the policy set loaded when it is built holds
rows only for the key paths some loaded constraint reads, and its ingest walks past sub-documents nothing reads -- syntax still
checked.  Same answers as the full table, fewer rows; a constraint that arrives later and reads another path makes the table
stale (GK_ERR_INVALID, "create it again"), as a new dictionary predicate does for every table.  (The whole CPU suite also passes
with GK_FORCE_PRUNE=1, which makes every table of every test a pruned one.)"""
import json

import numpy as np
import os

import pytest

from gatekeeper_amd import _lib as L
from gatekeeper_amd import driver as D
from gatekeeper_amd import synth
from parity_util import BACKENDS, make_client


def _load(c, fixtures, corpus=False):
    templates, constraints = synth.corpus(fixtures, 24) if corpus else (synth.psp_templates(fixtures), synth.audit_constraints())
    for t in templates:
        c.AddTemplate(t)
    for k in constraints:
        c.AddConstraint(k)


# (a run of the whole suite under GK_FORCE_PRUNE=1 -- every table pruned -- has no full table to compare with)
forced = pytest.mark.skipif(bool(os.environ.get("GK_FORCE_PRUNE")), reason="GK_FORCE_PRUNE=1: there is no unpruned table to compare with")


@forced
@pytest.mark.parametrize("backend", BACKENDS)
@pytest.mark.parametrize("corpus", [False, True], ids=["audit-50", "corpus-24"])
def test_pruned_table_answers_like_the_full_one(backend, fixtures, corpus):
    c = make_client(backend)
    _load(c, fixtures, corpus)
    eng = c.driver.engine
    n = 1500
    batch = synth.NativeBatch(eng.lib, n, seed=synth.SEED, mixed=True, start=0, namespaces=synth.gen_namespaces())
    full = eng.create_table_native(batch.reviews, n, resident=True, keep_text=True)
    lean = eng.create_table_native(batch.reviews, n, resident=True, keep_text=True, pruned=True)
    try:
        a, b = full.eval(want_match=True), lean.eval(want_match=True)
        assert (a.viol == b.viol).all() and (a.err == b.err).all() and (a.match == b.match).all() and (a.too_big == b.too_big).all()
        assert int(a.counts.sum()) > 500
        assert b.n_rows < 0.6 * a.n_rows and b.n_rows_read == a.n_rows_read          # fewer rows, the same rows READ
        assert a.algo_bytes == b.algo_bytes                                         # ... hence the same algorithmic bytes
        assert full.totals() == lean.totals()                                       # RESULT totals: counting plans + renderer (from the kept text)
        assert lean.stats()["device_bytes"] < full.stats()["device_bytes"]
    finally:
        full.free()
        lean.free()


@forced
def test_a_totals_plan_the_pruned_table_cannot_answer_leaves_the_whole_constraint_to_the_renderer(fixtures, monkeypatch):
    """A constraint's flag row and its count rows can sit in different totals plans.  When a pruned table lacks a path one of them
    reads, the constraint's pairs must ALL be rendered: a flag row answering "counted" next to a missing count row undercounted
    (K8sContainerLimits of the corpus: 645 results against 706).  The test aid makes the match formulas' label tests dictionary bits,
    which the frozen counting plans lower from label rows the pruned table does not hold."""
    c = make_client("hostemu")
    # (the label tests as dictionary bits of their OWN rows -- round 6 gives them to the review facts row, which lives in the main space
    #  and is answered for every plan: with it on, this situation cannot be reached any more)
    assert c.driver.engine.lib.gk_debug_set(b"fold_match_labels", 1) == 0 and c.driver.engine.lib.gk_debug_set(b"dict_facts", 0) == 0
    full = lean = None
    try:   # (the aids stay set while the plans are lowered -- at the first evaluation and the first totals --, not only while the policies load)
        _load(c, fixtures, True)
        eng = c.driver.engine
        n = 1500
        batch = synth.NativeBatch(eng.lib, n, seed=synth.SEED, mixed=True, start=0, namespaces=synth.gen_namespaces())
        full = eng.create_table_native(batch.reviews, n, resident=True, keep_text=True)
        lean = eng.create_table_native(batch.reviews, n, resident=True, keep_text=True, pruned=True)
        full.eval(), lean.eval()
        assert full.totals() == lean.totals()
        assert lean.rendered_pairs > 4 * full.rendered_pairs       # ... because the pruned table's unanswered constraints were rendered
    finally:
        c.driver.engine.lib.gk_debug_set(b"fold_match_labels", 0)
        c.driver.engine.lib.gk_debug_set(b"dict_facts", 1)
        for t in (full, lean):
            if t is not None:
                t.free()


@forced
def test_a_constraint_that_reads_another_path_makes_pruned_tables_stale(fixtures):
    c = make_client("hostemu")
    fx = fixtures
    t_priv = next(t for t in synth.psp_templates(fx) if t["spec"]["crd"]["spec"]["names"]["kind"] == "K8sPSPPrivilegedContainer")
    t_host = next(t for t in synth.psp_templates(fx) if t["spec"]["crd"]["spec"]["names"]["kind"] == "K8sPSPHostNamespace")
    c.AddTemplate(t_priv)
    c.AddTemplate(t_host)
    c.AddConstraint({"apiVersion": "constraints.gatekeeper.sh/v1beta1", "kind": "K8sPSPPrivilegedContainer", "metadata": {"name": "p"}, "spec": {}})
    eng = c.driver.engine
    batch = synth.NativeBatch(eng.lib, 200, seed=3, mixed=False, start=0, namespaces=synth.gen_namespaces())
    lean = eng.create_table_native(batch.reviews, 200, pruned=True)
    full = eng.create_table_native(batch.reviews, 200)
    lean.eval()
    c.AddConstraint({"apiVersion": "constraints.gatekeeper.sh/v1beta1", "kind": "K8sPSPHostNamespace", "metadata": {"name": "h"}, "spec": {}})   # reads spec.hostPID / hostIPC
    # (round 6: hostPID / hostIPC are tests on non-iterated leaves -- bits of the review facts row, i.e. new dictionary predicates: EVERY
    #  table flattened before is stale, the pruned one for two reasons; "create it again" either way)
    with pytest.raises(D.EngineError, match="GK_TABLE_PRUNED|create it again"):
        lean.eval()
    with pytest.raises(D.EngineError, match="create it again"):
        full.eval()
    full2 = eng.create_table_native(batch.reviews, 200)
    ev = full2.eval()                                                              # a full table built now serves the new policy set
    again = eng.create_table_native(batch.reviews, 200, pruned=True)                # ... and so does a pruned one
    assert (again.eval().viol == ev.viol).all() and int(ev.counts.sum()) > 0
    # with the review facts off the new constraint reads ROWS the pruned table lacks and the full table holds: the round-4 contract
    eng.lib.gk_debug_set(b"dict_facts", 0)
    try:
        c2 = make_client("hostemu")
        c2.AddTemplate(t_priv)
        c2.AddTemplate(t_host)
        c2.AddConstraint({"apiVersion": "constraints.gatekeeper.sh/v1beta1", "kind": "K8sPSPPrivilegedContainer", "metadata": {"name": "p"}, "spec": {}})
        e2 = c2.driver.engine
        lean2 = e2.create_table_native(batch.reviews, 200, pruned=True)
        full3 = e2.create_table_native(batch.reviews, 200)
        lean2.eval()
        c2.AddConstraint({"apiVersion": "constraints.gatekeeper.sh/v1beta1", "kind": "K8sPSPHostNamespace", "metadata": {"name": "h"}, "spec": {}})
        with pytest.raises(D.EngineError, match="GK_TABLE_PRUNED"):
            lean2.eval()
        assert (full3.eval().viol == ev.viol).all()                                 # the full table serves the new policy set
        lean2.free(); full3.free()
    finally:
        eng.lib.gk_debug_set(b"dict_facts", 1)
    for t in (lean, full, full2, again):
        t.free()


def test_what_the_pruned_ingest_walks_past_is_still_checked(fixtures):
    """malformed JSON inside a sub-document nothing reads: the review is rejected exactly as in a full table (the one-pass
    parser declines, the general path reports the decoder's error)"""
    c = make_client("hostemu")
    _load(c, fixtures)
    eng = c.driver.engine
    good = json.dumps({"apiVersion": "v1", "kind": "Pod", "metadata": {"name": "ok", "namespace": "prod-01"}, "status": {"conditions": [{"type": "Ready", "x": [1, 2, {"y": None}]}]},
                       "spec": {"containers": [{"name": "c", "image": "i", "securityContext": {"privileged": True}}]}})
    docs = [good, good.replace('"Ready"', '"Ready" "oops"'), good.replace("[1, 2,", "[1, 2,,"), good.replace('"x": [1', '"x": [01e'), good[:-1]]
    rins = [D.ReviewIn(L.GK_REVIEW_OBJECT, d.encode(), None, None, "Original", "") for d in docs]
    st = {}
    for pruned in (False, True):
        import os
        os.environ["GK_FORCE_PRUNE" if pruned else "GK_NO_PRUNE"] = "1"
        try:
            t = eng.create_table(rins, keep_docs=False)
            st[pruned] = (list(t.statuses), [int(x) for x in np.unpackbits(t.eval().viol.view(np.uint8), bitorder="little")[:8]])
            t.free()
        finally:
            os.environ.pop("GK_FORCE_PRUNE", None)
            os.environ.pop("GK_NO_PRUNE", None)
    assert st[True] == st[False] and st[True][0][0] == 0 and all(s != 0 for s in st[True][0][1:3]) and st[True][0][4] != 0
