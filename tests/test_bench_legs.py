"""bench.py's checker legs on the CPU build: the independent pure-Python oracle leg (oracle/bench_leg.py) must reproduce the
bitmaps of the table it is pointed at, from the batch's JSON text alone -- and must notice a flipped bit."""
import numpy as np

import bench
from gatekeeper_amd import driver as D
from gatekeeper_amd import synth


def test_python_oracle_leg_checks_the_bitmaps(fixtures):
    templates, constraints = synth.psp_templates(fixtures), synth.audit_constraints()
    drv = D.Driver(device=0, hostemu=True)
    client = D.Client(drv)
    for t in templates:
        client.AddTemplate(t)
    for k in constraints:
        client.AddConstraint(k)
    defaulted = [client.constraints[(k["kind"], k["metadata"]["name"])] for k in constraints]
    bench.batch_constraint_ids[:] = [drv.constraint_id(c) for c in defaulted]
    n = 320
    batch = synth.NativeBatch(drv.engine.lib, n, seed=synth.SEED, mixed=True, start=0, namespaces=synth.gen_namespaces())
    table = drv.engine.create_table_native(batch.reviews, n, keep_docs=False, resident=True)
    table.launch()
    ev = table.eval(download=True, collect_only=True)
    leg = bench.python_oracle_leg(templates, constraints, batch, ev, n)
    assert leg["pairs_equal"] and leg["n"] == n and leg["device_violating_pairs"] == leg["oracle_violating_pairs"] > 50
    ev.viol = np.array(ev.viol, copy=True)
    ev.viol[3][1] ^= np.uint64(1 << 17)          # one wrong bit must be found
    bad = bench.python_oracle_leg(templates, constraints, batch, ev, n)
    assert not bad["pairs_equal"] and bad["only_device"] + bad["only_oracle"] == 1


def test_result_totals_from_kept_text_equal_kept_docs_and_the_oracle(fixtures):
    """RESULT totals (pkg/audit/manager.go:902: one per types.Result) of a table built WITHOUT parsed documents
    (GK_TABLE_KEEP_TEXT: only the violating reviews are parsed, once each) = the totals of a GK_TABLE_KEEP_DOCS table = the
    oracle's result counts; gk_render works from the kept text as well."""
    from oracle import client as OC
    from oracle import target as OT
    templates, constraints = synth.psp_templates(fixtures), synth.audit_constraints()
    drv = D.Driver(device=0, hostemu=True)
    client, oc = D.Client(drv), OC.Client()
    for t in templates:
        client.AddTemplate(t)
        oc.add_template(t)
    for k in constraints:
        client.AddConstraint(k)
        oc.add_constraint(k)
    n = 400
    nss = synth.gen_namespaces()
    batch = synth.NativeBatch(drv.engine.lib, n, seed=synth.SEED, mixed=True, start=0, namespaces=nss)
    lean = drv.engine.create_table_native(batch.reviews, n, keep_docs=False, resident=True, keep_text=True)
    full = drv.engine.create_table_native(batch.reviews, n, keep_docs=True, resident=True)
    ev = lean.eval()
    full.eval()
    a, b = lean.totals(), full.totals()
    assert a == b and sum(r for r, _ in a.values()) > 100
    want = {}
    for o in synth.gen_objects(n, seed=synth.SEED, mixed=True):
        for r in oc.review(OT.AugmentedUnstructured(OT.Unstructured(o), synth.namespace_for(o, nss), "Original"), OC.AUDIT_EP):
            key = (r.constraint["kind"], r.constraint["metadata"]["name"])
            want[key] = want.get(key, 0) + 1
    got = {}
    for cid, (cons, _, _) in client._active(D.AUDIT_EP).items():
        if a.get(cid, (0, 0))[0]:
            got[(cons["kind"], cons["metadata"]["name"])] = a[cid][0]
    assert got == want
    row = {int(cid): i for i, cid in enumerate(ev.constraint_ids)}
    cid = next(c for c in a if a[c][1])
    r = int(D.EvalResult.bits(ev.viol[row[cid]], ev.n_reviews)[0])
    assert lean.render(cid, r) == full.render(cid, r) and lean.render(cid, r)
    lean.free()
    full.free()


def test_other_configs_legs_run_on_the_cpu_build(fixtures, monkeypatch):
    """The default bench line's `other_configs` (configs[1], configs[4] resident + streaming, each with its own parity leg against
    the pure-Python oracle at the leg's own table geometry): the Python plumbing of bench.side_point on the TEST-ONLY CPU build of
    the engine, small sizes -- so that a GPU visit is not spent on a typo.  The product bench never takes this route: bench.main
    asserts a device and side_point asks for hostemu=False; here Driver is patched."""
    import argparse
    import torch

    class EmuDriver(D.Driver):
        def __init__(self, device=0, hostemu=None, **kw):
            super().__init__(device=device, hostemu=True, **kw)
    monkeypatch.setattr(D, "Driver", EmuDriver)
    monkeypatch.setattr(torch.cuda, "synchronize", lambda *a, **k: None)
    nss = synth.gen_namespaces()
    out, stream = bench.side_point(1, 384, 2, 1, 128, 0, fixtures, nss)
    assert stream is None and out["parity_python_oracle"]["pairs_equal"] and out["parity_python_oracle"]["n"] == 128
    assert out["parity_compiled_independent"]["pairs_equal"] and out["parity_compiled_independent"]["n"] == 384
    assert out["plan_groups"] == 1 and out["roofline"]["algo_bytes_per_sweep_table_once"] == out["roofline"]["algo_bytes_per_sweep_every_group"] > 0
    sa = argparse.Namespace(stream_batches=2, warmup=1, stream_unique=2, batch=128, offered=0.0)
    from gatekeeper_amd import _lib
    lib = _lib.load(hostemu=True)
    lib.gk_debug_set(b"group_max", 64)     # (the several-group legs of the bench: the corpus is one plan since round 6)
    try:
        out, stream = bench.side_point(4, 256, 1, 0, 64, 0, fixtures, nss, with_stream=True, stream_args=sa, dev=None, totals=True, n_templates=140)
    finally:
        lib.gk_debug_set(b"group_max", 0)
    tl = out["audit_result_totals"]
    assert "error" not in tl and tl["host_pass_over_every_pair"]["equal"] and tl["results"] > tl["violating_pairs"] and tl["rendered_share"] < 0.1, tl
    assert out["plan_groups"] == 3 and out["parity_python_oracle"]["pairs_equal"], out.get("parity_python_oracle")
    assert out["parity_compiled_independent"].get("pairs_equal") and out["parity_compiled_independent"]["n"] == 256, out["parity_compiled_independent"]
    assert tl["independent_compiled_checker"]["equal"] and tl["independent_compiled_checker"]["checker_results"] == tl["results"], tl["independent_compiled_checker"]
    assert "_results_by_row" not in out["parity_compiled_independent"]
    pm = out["parity_messages_compiled_independent"]
    assert pm.get("messages_equal") and pm["objects"] == 256 and pm["messages"] >= pm["violating_pairs"] > 256, pm
    assert out["roofline"]["algo_bytes_per_sweep_table_once"] < out["roofline"]["algo_bytes_per_sweep_every_group"]
    assert "error" not in stream and stream["parity_python_oracle"]["pairs_equal"] and stream["batches"] == 2
    assert stream["parity_compiled_independent"].get("pairs_equal") and stream["parity_compiled_independent"]["n"] == 128, stream["parity_compiled_independent"]


def test_compiled_independent_leg_checks_the_bitmaps(fixtures):
    """bench.indep_leg: the independent compiled checker (oracle/libgkindep.so) over every object of the table it is pointed at, from
    the batch's JSON text alone; equal to the product's bitmaps, and a flipped bit is found (the table ends inside a bitmap word)."""
    templates, constraints = synth.psp_templates(fixtures), synth.audit_constraints()
    drv = D.Driver(device=0, hostemu=True)
    client = D.Client(drv)
    for t in templates:
        client.AddTemplate(t)
    for k in constraints:
        client.AddConstraint(k)
    defaulted = [client.constraints[(k["kind"], k["metadata"]["name"])] for k in constraints]
    bench.batch_constraint_ids[:] = [drv.constraint_id(c) for c in defaulted]
    n = 1000
    batch = synth.NativeBatch(drv.engine.lib, n, seed=synth.SEED, mixed=True, start=0, namespaces=synth.gen_namespaces())
    table = drv.engine.create_table_native(batch.reviews, n, keep_docs=False, resident=True, keep_text=True, pruned=True)
    table.launch()
    ev = table.eval(download=True, collect_only=True)
    base, par = bench.indep_leg(templates, constraints, batch, ev, budget_s=0.2)
    vs = bench.totals_against_checker(table, dict(par))      # (the main line's call: gk_table_totals against the checker's RESULT totals)
    assert vs["equal"] and vs["checker_results"] == vs["product_results"] == par["checker_results"]
    assert par["checker_results"] == sum(par["_results_by_row"]) >= par["checker_violating_pairs"]
    assert par["pairs_equal"] and par["n"] == n and par["device_violating_pairs"] == par["checker_violating_pairs"] > 500
    assert base["kind"] == "port" and base["cores"] == 1 and base["value"] > 0 and base["all_cores"]["sample_reviews"] == n
    ev.viol = np.array(ev.viol, copy=True)
    ev.viol[5][2] ^= np.uint64(1 << 9)
    _, bad = bench.indep_leg(templates, constraints, batch, ev, budget_s=0.1)
    assert not bad["pairs_equal"]


def test_messages_leg_compares_rendered_text_with_the_compiled_checker(fixtures):
    """bench.messages_leg: every message rendered for a prefix of the table against the compiled checker's text; a table whose bitmaps
    claim a pair the checker has no message for is reported with the first difference"""
    templates, constraints = synth.psp_templates(fixtures), synth.audit_constraints()
    drv = D.Driver(device=0, hostemu=True)
    client = D.Client(drv)
    for t in templates:
        client.AddTemplate(t)
    for k in constraints:
        client.AddConstraint(k)
    defaulted = [client.constraints[(k["kind"], k["metadata"]["name"])] for k in constraints]
    bench.batch_constraint_ids[:] = [drv.constraint_id(c) for c in defaulted]
    n = 600
    batch = synth.NativeBatch(drv.engine.lib, n, seed=synth.SEED, mixed=True, start=0, namespaces=synth.gen_namespaces())
    table = drv.engine.create_table_native(batch.reviews, n, keep_docs=False, resident=True, keep_text=True, pruned=True)
    table.launch()
    ev = table.eval(download=True, collect_only=True)
    leg = bench.messages_leg(templates, constraints, batch, ev, table, n_objects=400)
    assert leg["messages_equal"] and leg["objects"] == 400 and leg["messages"] >= leg["violating_pairs"] > 300, leg
    ev.viol = np.array(ev.viol, copy=True)
    row = next(r for r in range(len(constraints)) if not (int(ev.viol[r][0]) >> 7) & 1)
    ev.viol[row][0] |= np.uint64(1 << 7)      # a pair the policy does not produce: the product renders nothing for it, the checker has no entry
    bad = bench.messages_leg(templates, constraints, batch, ev, table, n_objects=64)
    assert not bad["messages_equal"] and bad["first_difference"]["object"] == 7


def test_strided_compiled_leg_covers_the_edges_and_finds_a_flipped_bit(fixtures):
    """bench.strided_indep_leg (the parity leg of the 10 M-object configs[3] table): whole bitmap words -- first, middle, last and an
    even spread -- re-evaluated by the independent compiled checker; equal to the product's bitmaps on a table that ends inside a word,
    the per-constraint totals of the sample agree, the prefix totals reproduce a fully evaluated smaller table, and one flipped bit in
    the LAST word is found."""
    templates, constraints = synth.psp_templates(fixtures), synth.audit_constraints()
    drv = D.Driver(device=0, hostemu=True)
    client = D.Client(drv)
    for t in templates:
        client.AddTemplate(t)
    for k in constraints:
        client.AddConstraint(k)
    defaulted = [client.constraints[(k["kind"], k["metadata"]["name"])] for k in constraints]
    ids = [drv.constraint_id(c) for c in defaulted]
    n = 2021
    nss = synth.gen_namespaces()
    batch = synth.NativeBatch(drv.engine.lib, n, seed=synth.SEED, mixed=True, start=0, namespaces=nss)
    table = drv.engine.create_table_native(batch.reviews, n, keep_docs=False, resident=True)
    table.launch()
    ev = table.eval(download=True, collect_only=True)
    words = bench.strided_words(n, 640, 2)
    assert list(words[:2]) == [0, 1] and list(words[-2:]) == [30, 31] and 15 in words and 10 <= len(words) <= 12
    # the fully evaluated table of the first 640 objects of the same stream: its per-constraint pairs are this table's prefix totals
    small = synth.NativeBatch(drv.engine.lib, 640, seed=synth.SEED, mixed=True, start=0, namespaces=nss)
    st = drv.engine.create_table_native(small.reviews, 640, keep_docs=False, resident=True)
    st.launch()
    sev = st.eval(download=True, collect_only=True)
    row = {int(c): i for i, c in enumerate(sev.constraint_ids)}
    known = (640, [int(sev.counts[row[c]]) for c in ids])
    leg = bench.strided_indep_leg(templates, constraints, batch, ev, ids=ids, want_objects=640, known_prefix=known, edge_words=2)
    assert leg["pairs_equal"] and leg["per_constraint_totals_equal"] and leg["first_difference"] is None, leg
    assert leg["last_word"] == leg["table_words"] - 1 == 31 and leg["first_word"] == 0 and leg["n"] == (len(words) - 1) * 64 + n % 64
    assert leg["device_violating_pairs"] == leg["checker_violating_pairs"] > 100 and leg["prefix_totals"]["equal"]
    ev.viol = np.array(ev.viol, copy=True)
    ev.viol[7][31] ^= np.uint64(1 << 3)           # review 1987, in the table's last (partial) word
    bad = bench.strided_indep_leg(templates, constraints, batch, ev, ids=ids, want_objects=640, edge_words=2)
    assert not bad["pairs_equal"] and bad["first_difference"]["table_word"] == 31
    # the whole table when it is small enough
    assert len(bench.strided_words(n, 1 << 20)) == 32
