#!/usr/bin/env python3
"""Host ingest alone (JSON text -> HandleReview -> rows), on the GPU-less test build: what one host thread does, where no GPU
is needed to know it.  `python tools/ingest_probe.py [--reviews 200000] [--threads 1] [--repeat 4] [--config 2|4]`
prints reviews/s and MB/s of JSON per pass (the first pass pays the thread's path tables and staging pool)."""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--reviews", type=int, default=200000)
    ap.add_argument("--threads", type=int, default=1)
    ap.add_argument("--repeat", type=int, default=4)
    ap.add_argument("--config", type=int, default=2, choices=[2, 4])
    ap.add_argument("--no-prune", action="store_true")
    ap.add_argument("--high-cardinality", action="store_true", help="every container's image tag and name unique in the stream: the per-value memos of the "
                                                                     "dictionary expressions never hit (the default vocabulary of a dozen images always does)")
    a = ap.parse_args()
    os.environ["GK_HOST_THREADS"] = str(a.threads)
    from gatekeeper_amd import driver as D
    from gatekeeper_amd import synth
    fx = synth.load_fixtures()
    if a.config == 2:
        templates, constraints = synth.psp_templates(fx), synth.audit_constraints()
    else:
        templates, constraints = synth.corpus(fx, 200)
    drv = D.Driver(device=0, hostemu=True)   # the test-only CPU build: ingest is the same host objects
    client = D.Client(drv)
    for t in templates:
        client.AddTemplate(t)
    for k in constraints:
        client.AddConstraint(k)
    batch = synth.NativeBatch(drv.engine.lib, a.reviews, seed=synth.SEED, mixed=True, start=0, namespaces=synth.gen_namespaces(), high_cardinality=a.high_cardinality)
    for i in range(a.repeat):
        t0 = time.perf_counter()
        table = drv.engine.create_table_native(batch.reviews, a.reviews, keep_docs=False, resident=True, pruned=not a.no_prune)
        dt = time.perf_counter() - t0
        st = table.stats()
        jb = st.get("json_bytes") or 0
        fl = st.get("flatten_s") or 0
        # (the whole call includes the CPU build's stand-in for the device-side assembly of the parts -- memcpy that a GPU box does not pay;
        #  `flatten only` is the host ingest proper: JSON text -> rows of every part)
        print("pass %d: %.3f s  %.0f reviews/s  %.0f MB/s of JSON  | flatten only %.3f s  %.0f reviews/s  %.0f MB/s  (rows %s)" % (
            i, dt, a.reviews / dt, jb / dt / 1e6, fl, a.reviews / fl if fl else 0, jb / fl / 1e6 if fl else 0, st.get("rows")))
        table.free()


if __name__ == "__main__":
    main()
