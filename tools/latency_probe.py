#!/usr/bin/env python3
"""Admission latency through the micro-batcher (row f1): T threads call Driver.Query concurrently on synthetic Pod reviews
against the 30 PSP constraints; per-call latency (arrival -> results, gk_query_stats.total_us) p50 / p99, batch sizes and
sustained reviews/s, for several concurrency levels and batching windows.  Prints one JSON object."""
import json
import os
import sys
import threading
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from gatekeeper_amd import driver as D   # noqa: E402
from gatekeeper_amd import synth         # noqa: E402


def run(drv, cons, reviews, threads, per_thread):
    lat, sizes = [], []
    lock = threading.Lock()

    def worker(w):
        mine, ms = [], []
        for k in range(per_thread):
            rv = reviews[(w * per_thread + k) % len(reviews)]
            t0 = time.perf_counter()
            drv.Query(D.TARGET_NAME, cons, rv)
            mine.append((time.perf_counter() - t0) * 1e6)
            ms.append(drv.last_query_stats["batch_size"])
        with lock:
            lat.extend(mine)
            sizes.extend(ms)

    ts = [threading.Thread(target=worker, args=(w,)) for w in range(threads)]
    t0 = time.perf_counter()
    for t in ts:
        t.start()
    for t in ts:
        t.join()
    dt = time.perf_counter() - t0
    lat.sort()
    return {"threads": threads, "calls": len(lat), "p50_us": lat[len(lat) // 2], "p99_us": lat[int(len(lat) * 0.99)], "max_us": lat[-1],
            "mean_batch": sum(sizes) / len(sizes), "reviews_per_s": len(lat) / dt}


def main():
    hostemu = len(sys.argv) > 1 and sys.argv[1] == "hostemu"
    fx = synth.load_fixtures()
    drv = D.Driver(hostemu=hostemu)
    client = D.Client(drv)
    for t in synth.psp_templates(fx):
        client.AddTemplate(t)
    for k in synth.psp_constraints():
        client.AddConstraint(k)
    cons = list(client.constraints.values())
    nss = synth.gen_namespaces()
    objs = synth.gen_objects(2048, seed=77)
    reviews = [D.AugmentedUnstructured(D.Unstructured(o), synth.namespace_for(o, nss), "Original") for o in objs]
    drv.Query(D.TARGET_NAME, cons, reviews[0])     # plan + kernel specialisation happen on the first call
    out = {"what": "Driver.Query through gk_query (30 PSP constraints, synthetic Pod reviews); latency includes the Python/ctypes call, "
                   "JSON encoding of the review, flatten, H2D, launch, D2H and rendering", "runs": []}
    for window in (0, 200, 1000):
        drv.StartBatcher(max_batch=64, window_us=window)
        for threads in (1, 16, 64):
            r = run(drv, cons, reviews, threads, 64 if threads > 1 else 256)
            r["window_us"] = window
            out["runs"].append(r)
    # the same path driven by NATIVE threads (gk_synth_query_storm, include/gksynth.h): no Python, no GIL -- what a cgo
    # shim's request goroutines would see.  Reviews are AugmentedUnstructured{Pod, Namespace} documents as JSON text.
    for shape in ("object", "admission_request"):
        batch = synth.NativeBatch(drv.engine.lib, 8192, seed=78, namespaces=nss, requests=(shape == "admission_request"))
        batch.query_storm(drv.engine, 8, 64)    # warm: dictionary growth + kernel specialisation for the batch geometries
        key = "native" if shape == "object" else "native_admission_request"
        out[key] = {"what": "gk_query from native threads (30 PSP constraints, synthetic Pod reviews as JSON text, shape: %s); latency = arrival -> "
                            "results ready (gk_query_stats.total_us): queueing + flatten + H2D + launch + D2H + rendering" % (
                                "AugmentedUnstructured{Pod, Namespace}" if shape == "object" else "admissionv1.AdmissionRequest (CREATE) + Namespace, the webhook's wire shape"),
                    "runs": []}
        combos = ((64, 0, 1), (64, 0, 2), (64, 100, 2), (256, 200, 2), (256, 200, 4), (1024, 500, 4), (1024, 500, 8)) if shape == "object" else ((64, 0, 2), (64, 100, 2), (256, 200, 4))
        for max_batch, window, workers in combos:
            drv.StartBatcher(max_batch=max_batch, window_us=window, workers=workers)
            for threads in (1, 8, 64, 256):
                per = max(64, min(2048, 16384 // threads))
                r = batch.query_storm(drv.engine, threads, per)
                r["window_us"], r["max_batch"], r["workers"] = window, max_batch, workers
                out[key]["runs"].append(r)
        if shape == "admission_request":
            # Driver.Query as the Go shim calls it (round 6): gk_query_ex2 with every loaded constraint's id, GK_QUERY_PRE_MATCHED --
            # beside the same storm through gk_query (the engine matches as well), same batcher settings
            ids = sorted(drv.constraint_id(c) for c in cons)
            out["native_pre_matched"] = {"what": "gk_query_ex2(constraint ids, GK_QUERY_PRE_MATCHED) from native threads, AdmissionRequest reviews: the reference's Driver.Query contract", "runs": []}
            for max_batch, window, workers in ((64, 0, 2), (256, 200, 4)):
                drv.StartBatcher(max_batch=max_batch, window_us=window, workers=workers)
                for threads in (1, 8, 64, 256):
                    per = max(64, min(2048, 16384 // threads))
                    for name, kw in (("engine_matches", {}), ("pre_matched", {"constraint_ids": ids, "pre_matched": True})):
                        r = batch.query_storm(drv.engine, threads, per, **kw)
                        r["window_us"], r["max_batch"], r["workers"], r["mode"] = window, max_batch, workers, name
                        out["native_pre_matched"]["runs"].append(r)
        batch.free()
    print(json.dumps(out))


if __name__ == "__main__":
    main()
