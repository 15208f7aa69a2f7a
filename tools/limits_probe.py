#!/usr/bin/env python3
"""What the engine's limits cost: reviews whose arrays exceed the LDS element capacity of the plan are
re-run by gk_eval_big (one wave per review, accumulators in HBM, atomicOr outputs); reviews beyond 255 iterated elements or
with an object where elements are iterated are refused (too_big -> the caller's CPU driver).  This probe times the big
variant: configs[2]'s policy set over N synthetic objects with the element capacities squeezed (non-resident table, so the
default capacities apply) so that a chosen share of the reviews overflows, and reports per share: reviews re-run, duration of
the dominant kernel, duration of the big variant, reviews refused.  Prints one JSON object."""
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from gatekeeper_amd import driver as D   # noqa: E402
from gatekeeper_amd import synth         # noqa: E402


def main():
    hostemu = len(sys.argv) > 1 and sys.argv[1] == "hostemu"
    n = 20000 if hostemu else 200000
    fx = synth.load_fixtures()
    out = {"what": "50 audit constraints x %d mixed synthetic objects, NON-resident table (default element capacities apply); "
                   "gk_eval_big re-runs the reviews whose arrays exceed them" % n, "runs": []}
    nss = synth.gen_namespaces()
    for caps in (None, (4, 8, 4), (3, 6, 3), (2, 4, 2), (1, 2, 1)):
        drv = D.Driver(device=0, hostemu=hostemu, **({"elem_cap": caps} if caps else {}))
        c = D.Client(drv)
        for t in synth.psp_templates(fx):
            c.AddTemplate(t)
        for k in synth.audit_constraints():
            c.AddConstraint(k)
        batch = synth.NativeBatch(drv.engine.lib, n, seed=synth.SEED, mixed=True, start=0, namespaces=nss)
        table = drv.engine.create_table_native(batch.reviews, n, keep_docs=False, resident=False)
        table.eval()                      # warm: plan upload, kernel build
        ev = table.eval()
        out["runs"].append({"elem_cap": list(caps) if caps else "default", "reviews": n, "reviews_rerun_by_gk_eval_big": int(ev.n_overflow),
                            "share": float(ev.n_overflow) / n, "dominant_kernel_ms": float(ev.fast_kernel_ms),
                            "big_variant_ms": float(ev.kernel_ms) - float(ev.fast_kernel_ms), "reviews_refused": len(ev.too_big_reviews()),
                            "violating_pairs": int(ev.counts.sum())})
        table.free()
        del c, drv
    pairs = {r["violating_pairs"] for r in out["runs"] if not r["reviews_refused"]}
    out["same_answers_at_every_capacity"] = len(pairs) == 1
    print(json.dumps(out))


if __name__ == "__main__":
    main()
