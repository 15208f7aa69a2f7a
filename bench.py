#!/usr/bin/env python3
"""Benchmark of the hot path: AdmissionReview x constraint evaluations per second (BASELINE.json `metric`).

A "step" = one pass of the hot path (Match + violation predicates for every loaded constraint) over one batch of
synthetic reviews that is already resident in HBM.  Default workload at N=1 = BASELINE.json configs[2], the largest
single-GPU configuration: the pkg/audit sweep, 50 constraints x 1 000 000 cached cluster objects on one MI355X
(--config 1 --reviews 100000 gives configs[1]: 30 PSP constraints x 100k Pod reviews).  With --gpus N (launched by
torch.distributed.run, one process per GPU) the objects are sharded: --scaling weak (default) gives every rank
`--reviews` objects, --scaling strong splits `--reviews` objects over the ranks (configs[3]: --reviews 10000000); the
per-shard violation bitmaps / counts are exchanged with one RCCL all-gather inside every step.

Prints ONE JSON line on rank 0 (contract in the task description) with
  roofline      dominant kernel: algorithmic bytes per launch / average launch duration (HIP events) vs 8 TB/s HBM
  end_to_end    the PCIe-inclusive leg: host parse + HandleReview + flatten + H2D of the same objects (never `value`)
  cpu_baseline  the independent compiled restatement of the reference's loop (oracle/indep_check.cpp) on 1 thread and on all host cores
  parity_sample / parity_compiled_independent   the device bitmaps of the timed table against that checker, every object, bit for bit
  parity_python_oracle   ... and against the pure-Python oracle (no code shared with the product either)
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0   # MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec


def indep_leg(templates, constraints, batch, ev, ids=None, budget_s=8.0, n=None):
    """The INDEPENDENT COMPILED checker (oracle/indep_check.cpp -> oracle/libgkindep.so: a C++ restatement of the Python oracle -- own
    JSON reader, value model, Rego parser + tree-walking interpreter, Match layer; no object of the product linked) over ALL objects
    of the timed table, taken as JSON text from the batch: its violation / autoreject bitmaps against the device's, bit for bit.
    -> (cpu_baseline of the restated reference: one thread on a bounded sample + all usable cores on the whole table, parity record)"""
    import numpy as np
    from oracle.indep_check import IndepChecker
    ck = IndepChecker(templates, constraints)
    nc, n = len(constraints), min(n or batch.n, batch.n)   # (n: the first n objects only -- a streamed batch of the same objects)
    cores = int(batch.lib.gk_host_cpus()) or os.cpu_count() or 1
    t0 = time.perf_counter()
    ck.check(batch.reviews, min(n, 512), 1)
    rate1 = min(n, 512) / max(time.perf_counter() - t0, 1e-9)
    n1 = int(max(64, min(n, rate1 * budget_s)) // 64 * 64) or n
    t0 = time.perf_counter()
    ck.check(batch.reviews, n1, 1)
    s1 = time.perf_counter() - t0
    t0 = time.perf_counter()
    viol, err, results = ck.check_totals(batch.reviews, n, cores)   # (the RESULT totals come out of the same pass)
    s_all = time.perf_counter() - t0
    ck.close()
    row_of = {int(cid): i for i, cid in enumerate(ev.constraint_ids)}
    words = (n + 63) // 64
    tail = np.uint64((1 << (n % 64)) - 1) if n % 64 else None
    equal, dev_pairs, ck_pairs, dev_err, ck_err = True, 0, 0, 0, 0
    for row, cid in enumerate(ids if ids is not None else batch_constraint_ids):
        d_v, d_e = np.array(ev.viol[row_of[cid]][:words], copy=True), np.array(ev.err[row_of[cid]][:words], copy=True)
        if tail is not None:
            d_v[-1] &= tail
            d_e[-1] &= tail
        equal = equal and bool((d_v == viol[row][:words]).all()) and bool((d_e == err[row][:words]).all())
        dev_pairs += int(np.unpackbits(d_v.view(np.uint8)).sum()); ck_pairs += int(np.unpackbits(viol[row][:words].view(np.uint8)).sum())
        dev_err += int(np.unpackbits(d_e.view(np.uint8)).sum()); ck_err += int(np.unpackbits(err[row][:words].view(np.uint8)).sum())
    base = {"value": n1 * nc / s1, "unit": "evals/s", "cores": 1, "kind": "port",
            "sample": "first %d of the same synthetic objects x %d constraints, %.1f s on 1 thread of the independent compiled restatement of the reference's "
                      "path (oracle/indep_check.cpp: per object JSON decode -> HandleReview -> per constraint match.Matches -> tree-walking Rego evaluation of the "
                      "template; own reader / parser / interpreter / Match layer, nothing of the product linked; the Go/OPA reference itself cannot be built here)" % (n1, nc, s1),
            "all_cores": {"value": n * nc / s_all, "cores": cores, "hardware_threads": os.cpu_count(), "sample_reviews": n, "seconds": s_all,
                          "note": "cores = CPUs usable under the affinity mask / cgroup CPU quota (gk_host_cpus), one thread each; the whole timed table"}}
    parity = {"n": n, "constraints": nc, "pairs_equal": equal, "device_violating_pairs": dev_pairs, "checker_violating_pairs": ck_pairs,
              "device_autoreject_pairs": dev_err, "checker_autoreject_pairs": ck_err, "seconds": s_all, "threads": cores,
              "checker": "oracle/indep_check.cpp -> oracle/libgkindep.so: compiled, independent of the product (one source file + the C++ standard library on its link line), "
                         "pinned against the Python oracle by tests/test_indep_check.py; every object of the timed table, bit for bit",
              "checker_results": int(results.sum()), "_results_by_row": [int(x) for x in results]}
    return base, parity


def strided_words(n, want_objects, edge_words=2048):
    """bitmap words (64 objects each) of an n-object table to re-evaluate: the first, the middle and the last `edge_words` words --
    the table's first, middle and LAST row groups, where 32-bit row / heap offsets and tile indices are largest -- plus evenly
    spread words in between until about `want_objects` objects are covered.  Sorted, distinct."""
    import numpy as np
    words = (n + 63) // 64
    if words * 64 <= want_objects or words <= 3 * edge_words:
        return np.arange(words, dtype=np.int64)
    mid = words // 2 - edge_words // 2
    fixed = np.concatenate([np.arange(edge_words), np.arange(mid, mid + edge_words), np.arange(words - edge_words, words)])
    rest = max(0, want_objects // 64 - len(fixed))
    spread = np.linspace(edge_words, words - edge_words - 1, num=rest, dtype=np.int64) if rest else np.zeros(0, np.int64)
    return np.unique(np.concatenate([fixed, spread]).astype(np.int64))


def strided_indep_leg(templates, constraints, batch, ev, ids=None, want_objects=1 << 20, known_prefix=None, edge_words=2048):
    """The independent compiled checker (oracle/libgkindep.so) over a STRIDED sample of a table too large to re-evaluate whole in
    the default run (configs[3]'s N = 1 point: 10 M objects): whole 64-object bitmap words -- the first, middle and last 2048 words
    and an even spread between them -- are re-evaluated from the batch's JSON text and compared, word for word, with the device's
    violation and autoreject bitmaps; per constraint the pair totals over the sample must agree as well.  known_prefix: (n, [pairs
    per constraint]) of a table of the first n objects of the same stream checked in full elsewhere (configs[2]): the device's
    popcounts over the first n objects of THIS table must reproduce them.
    Reference loop being restated: /root/reference/pkg/audit/manager.go:591-642."""
    import ctypes as C
    import numpy as np
    from gatekeeper_amd import _lib as L
    from oracle.indep_check import IndepChecker
    n = batch.n
    sel = strided_words(n, want_objects, edge_words)
    m = 0
    src = batch.reviews
    arr = (L.gk_review_in * (len(sel) * 64))()
    sz = C.sizeof(L.gk_review_in)
    base = C.addressof(src.contents)
    spans = []            # (first sample object, first table word, words) of every run of consecutive words
    run0 = 0
    for i in range(1, len(sel) + 1):
        if i == len(sel) or sel[i] != sel[i - 1] + 1:
            w0, nw = int(sel[run0]), i - run0
            cnt = min(nw * 64, n - w0 * 64)
            C.memmove(C.addressof(arr) + m * sz, base + w0 * 64 * sz, cnt * sz)
            spans.append((m, w0, nw))
            m += nw * 64 if cnt == nw * 64 else cnt
            run0 = i
    cores = int(batch.lib.gk_host_cpus()) or os.cpu_count() or 1
    ck = IndepChecker(templates, constraints)
    t0 = time.perf_counter()
    viol, err, results = ck.check_totals(arr, m, cores)
    seconds = time.perf_counter() - t0
    ck.close()
    row_of = {int(cid): i for i, cid in enumerate(ev.constraint_ids)}
    cids = list(ids if ids is not None else batch_constraint_ids)
    equal, dev_pairs, ck_pairs, dev_err, ck_err, totals_equal = True, 0, 0, 0, 0, True
    first_bad = None
    tail_word = (n // 64) if n % 64 else -1
    tail = np.uint64((1 << (n % 64)) - 1)
    for row, cid in enumerate(cids):
        dv_all, de_all = ev.viol[row_of[cid]], ev.err[row_of[cid]]
        d_v, d_e = np.array(dv_all[sel], copy=True), np.array(de_all[sel], copy=True)
        if tail_word >= 0 and sel[-1] == tail_word:
            d_v[-1] &= tail
            d_e[-1] &= tail
        c_v, c_e = viol[row][:len(sel)], err[row][:len(sel)]
        same = bool((d_v == c_v).all()) and bool((d_e == c_e).all())
        if not same and first_bad is None:
            k = int(np.nonzero((d_v != c_v) | (d_e != c_e))[0][0])
            first_bad = {"constraint": "%s/%s" % (constraints[row]["kind"], constraints[row]["metadata"]["name"]), "table_word": int(sel[k]),
                         "device": "%016x" % int(d_v[k]), "checker": "%016x" % int(c_v[k])}
        equal = equal and same
        a, b = int(np.unpackbits(d_v.view(np.uint8)).sum()), int(np.unpackbits(c_v.view(np.uint8)).sum())
        totals_equal = totals_equal and a == b
        dev_pairs += a; ck_pairs += b
        dev_err += int(np.unpackbits(d_e.view(np.uint8)).sum()); ck_err += int(np.unpackbits(c_e.view(np.uint8)).sum())
    out = {"n": int(m), "of": int(n), "words": int(len(sel)), "runs": len(spans), "first_word": int(sel[0]), "last_word": int(sel[-1]), "table_words": (n + 63) // 64,
           "constraints": len(cids), "pairs_equal": equal, "per_constraint_totals_equal": totals_equal,
           "device_violating_pairs": dev_pairs, "checker_violating_pairs": ck_pairs, "device_autoreject_pairs": dev_err, "checker_autoreject_pairs": ck_err,
           "checker_results": int(results.sum()), "seconds": seconds, "threads": cores, "first_difference": first_bad,
           "checker": "oracle/indep_check.cpp -> oracle/libgkindep.so (independent of the product) over a strided sample: the first, middle and last 2048 bitmap "
                      "words of the table and an even spread between them, every object of those words, bit for bit, plus the per-constraint pair totals of the sample"}
    if known_prefix is not None and known_prefix[0] <= n and known_prefix[0] % 64 == 0:
        pw = known_prefix[0] // 64
        mine = [int(np.unpackbits(np.ascontiguousarray(ev.viol[row_of[cid]][:pw]).view(np.uint8)).sum()) for cid in cids]
        out["prefix_totals"] = {"objects": int(known_prefix[0]), "equal": mine == [int(x) for x in known_prefix[1]], "pairs": int(sum(mine)),
                                "what": "device popcounts per constraint over the first %d objects of this table against the fully checked table of the same objects" % known_prefix[0]}
    return out


def totals_against_checker(table, parity, ids=None):
    """gk_table_totals of the timed table (RESULTS per constraint: device counts + host rendering of the flagged pairs) against the
    RESULT totals the independent compiled checker counted in its pass over the same objects (indep_leg).  parity: indep_leg's record;
    its private per-row list is consumed here."""
    by_row = parity.pop("_results_by_row", None)
    if by_row is None or parity.get("n") != table.n:
        return None
    tot = table.totals()
    cids = list(ids if ids is not None else batch_constraint_ids)
    equal = all(int(tot[cid][0]) == by_row[row] for row, cid in enumerate(cids))
    return {"equal": equal, "checker_results": int(sum(by_row)), "product_results": int(sum(r for r, _ in tot.values())),
            "what": "RESULT totals per constraint: gk_table_totals against the independent compiled checker's count of distinct (msg, details) per violating pair, every object of the table"}


def messages_leg(templates, constraints, batch, ev, table, ids=None, n_objects=2048):
    """Message TEXT against the independent compiled checker: for the first n_objects of the timed table every violating pair the device
    flagged is rendered by the product (gk_render over the kept JSON text) and compared, message for message, with the checker's own
    evaluation and formatting of the same object (oracle/indep_check.cpp ic_messages: its own number printing and sprintf)."""
    import numpy as np
    from oracle.indep_check import IndepChecker
    n = min(n_objects, batch.n)
    ck = IndepChecker(templates, constraints)
    cids = list(ids if ids is not None else batch_constraint_ids)
    row_of = {int(cid): i for i, cid in enumerate(ev.constraint_ids)}
    words = (n + 63) // 64
    flagged = [np.unpackbits(np.array(ev.viol[row_of[cid]][:words], copy=True).view(np.uint8), bitorder="little")[:n] for cid in cids]
    t0 = time.perf_counter()
    pairs = messages = 0
    first_difference = None
    for i in range(n):
        want = {row: sorted(m) for row, m in ck.messages(batch.json_text(i), batch.namespace_text(i)).items()}
        got = {row: sorted(v["msg"] for v in table.render(cid, i)) for row, cid in enumerate(cids) if flagged[row][i]}
        pairs += len(got)
        messages += sum(len(m) for m in got.values())
        if got != want and first_difference is None:
            row = next(r for r in sorted(set(got) | set(want)) if got.get(r) != want.get(r))
            first_difference = {"object": i, "constraint": "%s/%s" % (constraints[row]["kind"], constraints[row]["metadata"]["name"]),
                                "product": got.get(row), "checker": want.get(row)}
    ck.close()
    return {"objects": n, "violating_pairs": pairs, "messages": messages, "messages_equal": first_difference is None, "first_difference": first_difference,
            "seconds": time.perf_counter() - t0,
            "what": "every message the product renders for the first %d objects of the timed table (pairs flagged on the device) against the text the "
                    "independent compiled checker produces for the same objects" % n}


def oracle_pairs_of(templates, constraints, batch, n):
    """(viol pairs, err pairs, seconds, processes) of the pure-Python oracle over the first n objects of `batch` (JSON text)"""
    from oracle import bench_leg as BL
    texts = [(batch.json_text(i), batch.namespace_text(i)) for i in range(n)]
    return BL.python_oracle_pairs(templates, constraints, texts)


def python_oracle_leg(templates, constraints, batch, ev, n=16384, ids=None, oracle=None):
    """The INDEPENDENT full-size parity leg: the pure-Python oracle (oracle/client.py -- its own JSON reader, HandleReview,
    Match layer and Rego interpreter, no code shared with the product) evaluates the first `n` of the timed objects, taken as
    JSON text from the batch, and its (constraint, object) pair sets must equal the device's violation and autoreject bitmaps."""
    import numpy as np
    n = min(n, batch.n)
    n = n // 64 * 64 or n
    viol, err, seconds, procs = oracle if oracle is not None else oracle_pairs_of(templates, constraints, batch, n)
    row_of = {int(cid): i for i, cid in enumerate(ev.constraint_ids)}
    words = (n + 63) // 64

    def dev_pairs(bm):
        out = set()
        for row, cid in enumerate(ids if ids is not None else batch_constraint_ids):
            bits = np.unpackbits(bm[row_of[cid]][:words].view(np.uint8), bitorder="little")[:n]
            out.update((row, int(i)) for i in np.nonzero(bits)[0])
        return out
    dv, de = dev_pairs(ev.viol), dev_pairs(ev.err)
    return {"n": n, "constraints": len(constraints), "pairs_equal": dv == viol and de == err, "device_violating_pairs": len(dv),
            "oracle_violating_pairs": len(viol), "device_autoreject_pairs": len(de), "oracle_autoreject_pairs": len(err),
            "only_device": len(dv - viol) + len(de - err), "only_oracle": len(viol - dv) + len(err - de),
            "seconds": seconds, "processes": procs, "oracle_evals_per_s": n * len(constraints) / seconds if seconds > 0 else None,
            "checker": "oracle/client.py: pure-Python restatement (json.loads of the batch's JSON text -> HandleReview -> match.Matches -> "
                       "tree-walking Rego interpreter); shares no code with the product"}


def stream_leg(args, drv, client, templates, constraints, nss, rank, world, dev, dist, keep_first=None):
    """configs[4]: STREAMING admission -- the offered load arrives in batches of `--batch` reviews (JSON text); a rank takes the
    batches  k = rank (mod world)  (round robin over the GPUs, no collective) and runs each through
        ingest (JSON -> rows on the host threads) -> H2D + device assembly -> launch(es) -> D2H of the bitmaps
    DOUBLE-BUFFERED: batch k+1 is ingested and uploaded (its own stream) while batch k is evaluated and downloaded (its own
    stream) by a second host thread.  Arrivals are open-loop at `--offered` reviews/s over all ranks; a batch's latency is
    completion - arrival (queueing included when the pipeline is slower than the offered load).
    Reference shape this replaces: one Client.Review per request goroutine, /root/reference/pkg/webhook/policy.go:142-146,826."""
    import queue
    import threading
    import numpy as np
    import torch
    from gatekeeper_amd import synth
    nb, bsz = args.stream_batches, args.batch
    mine = [k for k in range(nb + args.warmup) if k % world == rank]
    # distinct batches of the global synthetic stream, generated up front (the generator is not part of the path)
    uniq = min(len(mine), args.stream_unique)
    t_gen = time.perf_counter()
    batches = [synth.NativeBatch(drv.engine.lib, bsz, seed=synth.SEED, mixed=True, start=(rank + j * world) * bsz, namespaces=nss) for j in range(uniq)]
    t_gen = time.perf_counter() - t_gen
    nc = len(constraints)
    period = bsz / float(args.offered) if args.offered > 0 else 0.0     # seconds between arrivals over ALL ranks
    q = queue.Queue(maxsize=1)                                       # one table in flight behind the one being ingested
    done, errors = [], []

    def consumer():
        try:
            while True:
                item = q.get()
                if item is None:
                    return
                k, table, arrival, t_ingest0, t_ingest1 = item
                t_dev0 = time.perf_counter()
                ev = table.eval(download=True)
                t_dev1 = time.perf_counter()
                st = table.stats()
                if k == 0 and keep_first is not None:      # (objects [0, batch) of the global stream: the parity leg's objects)
                    keep_first.append(ev)
                done.append({"k": k, "arrival": arrival, "ingest_s": t_ingest1 - t_ingest0, "flatten_s": st["flatten_s"], "h2d_s": st["upload_s"],
                             "device_s": t_dev1 - t_dev0, "kernel_ms": float(ev.kernel_ms), "fast_kernel_ms": float(ev.fast_kernel_ms),
                             "complete": t_dev1, "pairs": int(ev.counts.sum()), "too_big": len(ev.too_big_reviews()),
                             "algo_bytes": int(ev.algo_bytes), "lds": int(ev.lds_bytes), "rows": int(ev.n_rows), "rows_read": int(ev.n_rows_read)})
                table.free()
        except Exception as ex:   # noqa: BLE001
            errors.append(ex)
            while q.get() is not None:
                pass

    th = threading.Thread(target=consumer)
    th.start()
    if dist is not None:
        dist.barrier()
    torch.cuda.synchronize()
    t0 = None
    for j, k in enumerate(mine):
        if j == (args.warmup + world - 1) // world:       # the first batches (plan upload, kernel builds, pool growth) are warm-up
            while not q.empty():
                time.sleep(0.0005)
            if dist is not None:
                dist.barrier()
            torch.cuda.synchronize()
            t0 = time.perf_counter()
        arrival = (t0 + ((k - args.warmup) // world) * period * world) if (t0 is not None and period > 0) else time.perf_counter()
        now = time.perf_counter()
        if arrival > now:
            time.sleep(arrival - now)
        else:
            arrival = arrival if period > 0 and t0 is not None else now
        b = batches[j % uniq]
        t_i0 = time.perf_counter()
        table = drv.engine.create_table_native(b.reviews, bsz, keep_docs=False, resident=True, pruned=not getattr(args, "no_prune", False))
        t_i1 = time.perf_counter()
        q.put((k, table, arrival, t_i0, t_i1))
        if errors:
            break
    q.put(None)
    th.join()
    if errors:
        raise errors[0]
    torch.cuda.synchronize()
    if dist is not None:
        dist.barrier()
    t1 = time.perf_counter()
    timed = [d for d in done if d["arrival"] >= (t0 or 0) - 1e-9][-(len(mine) - (args.warmup + world - 1) // world):]
    dt = t1 - t0
    if dist is not None:
        tmax = torch.tensor([dt], device=dev, dtype=torch.float64)
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        dt = float(tmax.item())
        cnt = torch.tensor([len(timed)], device=dev, dtype=torch.float64)
        dist.all_reduce(cnt)
        n_timed_all = int(cnt.item())
    else:
        n_timed_all = len(timed)
    lat = np.array([d["complete"] - d["arrival"] for d in timed]) * 1e3
    mean = lambda key: float(np.mean([d[key] for d in timed])) if timed else 0.0
    kernel_s = mean("fast_kernel_ms") / 1e3
    algo = int(np.mean([d["algo_bytes"] for d in timed])) if timed else 0
    return {"dt": dt, "batches_all": n_timed_all, "batches_rank0": len(timed), "reviews_all": n_timed_all * bsz, "t_gen": t_gen,
            "lat_ms": {"p50": float(np.percentile(lat, 50)), "p90": float(np.percentile(lat, 90)), "p99": float(np.percentile(lat, 99)), "max": float(lat.max()), "mean": float(lat.mean())},
            "stage_mean_ms": {"ingest_flatten_h2d": mean("ingest_s") * 1e3, "flatten": mean("flatten_s") * 1e3, "h2d": mean("h2d_s") * 1e3,
                              "launch_to_bitmaps": mean("device_s") * 1e3, "device_kernels": mean("kernel_ms")},
            "kernel_s": kernel_s, "algo_bytes": algo, "lds": timed[-1]["lds"] if timed else 0, "pairs_per_batch": mean("pairs"), "too_big": int(sum(d["too_big"] for d in timed)),
            "rows_per_batch": mean("rows"), "rows_read_per_batch": mean("rows_read"), "unique_batches": uniq}


batch_constraint_ids = []


def C_u64():
    import ctypes
    return ctypes.c_uint64()


def _sig(x, digits=4):
    return float("%.*g" % (digits, x)) if x is not None else None


def totals_leg(table):
    """audit RESULT totals of an evaluated table (gk_table_totals: pkg/audit/manager.go:893-904 counts types.Results): the device
    COUNTS the results of the violating pairs it can (round 4: thresholds per counted iteration) and flags the rest for the host
    renderer; checked against the host pass that renders EVERY violating pair (GK_TOTALS_RENDER_ALL=1)."""
    t_cold = time.perf_counter()
    tot = table.totals()                       # first call: prepares the counting forms, builds + uploads the totals plans
    t_cold = time.perf_counter() - t_cold
    t_tot = time.perf_counter()
    tot = table.totals()
    t_tot = time.perf_counter() - t_tot
    rendered = table.rendered_pairs
    os.environ["GK_TOTALS_RENDER_ALL"] = "1"
    try:
        t_all = time.perf_counter()
        tot_all = table.totals()
        t_all = time.perf_counter() - t_all
    finally:
        del os.environ["GK_TOTALS_RENDER_ALL"]
    pairs = int(sum(p for _, p in tot.values()))
    return {"seconds": t_tot, "seconds_first_call": t_cold, "results": int(sum(r for r, _ in tot.values())), "violating_pairs": pairs, "rendered_pairs": int(rendered),
            "rendered_share": rendered / pairs if pairs else 0.0, "host_pass_over_every_pair": {"seconds": t_all, "equal": tot == tot_all},
            "what": "gk_table_totals on the timed table (GK_TABLE_KEEP_TEXT): the device counts the results of the violating pairs it can tell apart by their "
                    "messages' heads and flags the others, the host renders those only; checked against rendering every violating pair"}


def side_point(config, reviews, steps, warmup, oracle_n, dev_index, fx, nss, with_stream=False, stream_args=None, dev=None, totals=False, n_templates=200,
               strided_n=0, known_prefix=None, warm_probe=False):
    """One more BASELINE config measured the way the headline one is -- its own engine, policy set and resident table --
    for the `other_configs` of the default bench line: `steps` sweeps of the table in HBM between two synchronisations,
    the dominant kernel's duration from per-launch HIP events, and (oracle_n > 0) the INDEPENDENT parity leg: the device
    bitmaps of THAT table (the bench's row-group geometry, plan groups and streams) against the pure-Python oracle on the
    first `oracle_n` objects, taken as JSON text.  with_stream: the same engine then runs configs[4]'s STREAM (stream_leg)
    and the bitmaps of its first batch -- the same objects -- are checked against the same oracle pairs.
    Reference shape being replaced: the serial audit loop, /root/reference/pkg/audit/manager.go:591-642."""
    import torch
    from gatekeeper_amd import driver as D
    from gatekeeper_amd import synth
    templates = synth.psp_templates(fx)
    constraints = synth.psp_constraints() if config == 1 else synth.audit_constraints()
    if config == 4:
        templates, constraints = synth.corpus(fx, n_templates)   # (200: configs[4]; the CPU test of this function loads fewer)
    t_pol = time.perf_counter()
    drv = D.Driver(device=dev_index, hostemu=False)
    client = D.Client(drv)
    for t in templates:
        client.AddTemplate(t)
    for k in constraints:
        client.AddConstraint(k)
    t_pol = time.perf_counter() - t_pol
    defaulted = [client.constraints[(k["kind"], k["metadata"]["name"])] for k in constraints]
    ids = [drv.constraint_id(c) for c in defaulted]
    nc = len(constraints)
    t_gen = time.perf_counter()
    batch = synth.NativeBatch(drv.engine.lib, reviews, seed=synth.SEED, mixed=(config != 1), start=0, namespaces=nss)
    t_gen = time.perf_counter() - t_gen
    table = drv.engine.create_table_native(batch.reviews, reviews, keep_docs=False, resident=True, keep_text=totals, pruned=True)
    st = table.stats()

    def local(k, download=False, time_each=False, kernel_only=False):
        for _ in range(k):
            table.launch(time_each=time_each, kernel_only=kernel_only)
        return table.eval(download=download, collect_only=True)
    t_first = time.perf_counter()
    local(1)                                   # (plan upload, hiprtc builds of every plan group, binding)
    t_first = time.perf_counter() - t_first
    if warmup:
        local(warmup)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    res = local(steps)
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    iso = local(min(steps, 20), time_each=True)   # an event pair per launch: the kernel's own duration
    b2b = local(min(steps, 20), kernel_only=True)  # the dominant kernel alone, back to back under ONE event pair (see main())
    final = local(1, download=True)
    groups = int(final.n_plan_groups)
    # several plan groups run on their own streams and overlap: the sum of their kernels' durations is not the sweep's duration --
    # the sweep is what the timed region measures; one group: the kernel's own events
    kernel_s = (b2b.fast_kernel_ms if b2b.fast_kernel_ms > 0 else iso.fast_kernel_ms) / 1e3 if groups == 1 else dt / steps
    once = int(final.algo_bytes_once)
    out = {"workload": "configs[%d]: %d constraints x %d synthetic %s, resident in HBM" % (config if config != 2 else 3 if reviews > 2000000 else 2, nc, reviews,
                                                                                               "Pod reviews" if config == 1 else "mixed cluster objects"),
           "constraints": nc, "reviews": reviews, "steps": steps, "ms_per_step": dt / steps * 1e3, "evals_per_s": nc * reviews * steps / dt,
           "plan_groups": groups, "rows": int(final.n_rows), "rows_read": int(final.n_rows_read), "table_bytes": int(st["device_bytes"]),
           "roofline": {"bound": "hbm", "kernel": "gk_jit_tiles", "algo_bytes_per_sweep_table_once": once, "algo_bytes_per_sweep_every_group": int(final.algo_bytes),
                        "seconds": kernel_s, "clock": "HIP events: one pair around consecutive launches of the kernel alone" if groups == 1 else "wall clock of the timed sweeps (the plan groups overlap on their streams)",
                        "sum_of_group_kernels_ms": iso.fast_kernel_ms, "achieved": once / kernel_s / 1e9, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                        "frac": once / kernel_s / 1e9 / HBM_PEAK_GBS, "lds_bytes_per_tile": int(final.lds_bytes)},
           "ingest": {"flatten_s": st["flatten_s"], "h2d_s": st["upload_s"], "json_bytes": st["json_bytes"], "generate_s": t_gen,
                      "reviews_per_s": reviews / (st["flatten_s"] + st["upload_s"])},
           "policy_load_s": t_pol, "first_sweep_s": t_first, "violating_pairs": int(final.counts.sum()), "reviews_beyond_limits": len(final.too_big_reviews()),
           "reviews_evaluated_on_host": len(getattr(final, "host_evaluated", []))}
    oracle = None
    if strided_n > 0:   # a table too large for the whole-table legs: the compiled checker over a strided sample (first / middle / last row groups included)
        try:
            out["parity_compiled_independent"] = strided_indep_leg(templates, constraints, batch, final, ids=ids, want_objects=strided_n, known_prefix=known_prefix)
        except Exception as ex:   # noqa: BLE001
            out["parity_compiled_independent"] = {"error": "%s: %s" % (type(ex).__name__, ex)}
    if oracle_n > 0:
        try:
            n = min(oracle_n, reviews) // 64 * 64
            oracle = oracle_pairs_of(templates, constraints, batch, n)
            out["parity_python_oracle"] = python_oracle_leg(templates, constraints, batch, final, n, ids=ids, oracle=oracle)
            out["parity_python_oracle"]["geometry"] = "the timed table itself: %d reviews, %d plan group(s), LDS %d B per row group" % (reviews, groups, int(final.lds_bytes))
        except Exception as ex:   # noqa: BLE001
            out["parity_python_oracle"] = {"error": "%s: %s" % (type(ex).__name__, ex)}
        try:   # ... and the independent COMPILED checker over EVERY object of this table (the 10 M-object point passes oracle_n = 0: no checker there)
            _, out["parity_compiled_independent"] = indep_leg(templates, constraints, batch, final, ids=ids, budget_s=0.5)
        except Exception as ex:   # noqa: BLE001
            out["parity_compiled_independent"] = {"error": "%s: %s" % (type(ex).__name__, ex)}
        if totals:   # (the table keeps the JSON text: its pairs can be rendered)
            try:
                out["parity_messages_compiled_independent"] = messages_leg(templates, constraints, batch, final, table, ids=ids, n_objects=1024)
            except Exception as ex:   # noqa: BLE001
                out["parity_messages_compiled_independent"] = {"error": "%s: %s" % (type(ex).__name__, ex)}
    if totals:
        try:
            out["audit_result_totals"] = totals_leg(table)
        except Exception as ex:   # noqa: BLE001
            out["audit_result_totals"] = {"error": "%s: %s" % (type(ex).__name__, ex)}
        try:   # ... against the RESULT totals the compiled checker counted over the same objects
            vs = totals_against_checker(table, out.get("parity_compiled_independent") or {}, ids=ids)
            if vs is not None:
                out["audit_result_totals"]["independent_compiled_checker"] = vs
        except Exception as ex:   # noqa: BLE001
            out["audit_result_totals"]["independent_compiled_checker"] = {"error": "%s: %s" % (type(ex).__name__, ex)}
    (out.get("parity_compiled_independent") or {}).pop("_results_by_row", None)
    table.free()
    if warm_probe:
        # what a RESTARTED process pays for its first sweep: the code objects held in memory are dropped, a second engine loads the same
        # policies and builds the same table -- its plan-specialised kernels come from the disk cache (on by default), not from hiprtc
        try:
            lib = drv.engine.lib
            h0, c0 = C_u64(), C_u64()
            lib.gk_jit_cache_stats(h0, c0)
            lib.gk_jit_cache_drop_memory()
            drv2 = D.Driver(device=dev_index, hostemu=False)
            client2 = D.Client(drv2)
            for t in templates:
                client2.AddTemplate(t)
            for k in constraints:
                client2.AddConstraint(k)
            table2 = drv2.engine.create_table_native(batch.reviews, reviews, keep_docs=False, resident=True, pruned=True)
            t_warm = time.perf_counter()
            table2.launch()
            again = table2.eval(download=True, collect_only=True)
            t_warm = time.perf_counter() - t_warm
            h1, c1 = C_u64(), C_u64()
            lib.gk_jit_cache_stats(h1, c1)
            out["first_sweep_warm_s"] = t_warm
            out["jit_cache"] = {"dir": (lib.gk_jit_cache_dir() or b"").decode(), "compiles_cold": int(c0.value), "compiles_during_warm_sweep": int(c1.value - c0.value),
                                "served_from_cache_during_warm_sweep": int(h1.value - h0.value), "same_pairs": int(again.counts.sum()) == out["violating_pairs"],
                                "what": "first_sweep_s: a fresh process, empty cache (hiprtc of every plan group, side by side); first_sweep_warm_s: a second engine after "
                                        "gk_jit_cache_drop_memory -- the code objects come from the disk cache"}
            table2.free()
            drv2.engine.close()
        except Exception as ex:   # noqa: BLE001
            out["jit_cache"] = {"error": "%s: %s" % (type(ex).__name__, ex)}
    stream = None
    if with_stream:
        try:
            first = []
            r = stream_leg(stream_args, drv, client, templates, constraints, nss, 0, 1, dev, None, keep_first=first)
            stream = {"offered_reviews_per_s": stream_args.offered, "batch": stream_args.batch, "batches": r["batches_all"], "achieved_reviews_per_s": r["reviews_all"] / r["dt"],
                      "evals_per_s": nc * r["reviews_all"] / r["dt"], "batch_latency_ms": r["lat_ms"], "stage_mean_ms": r["stage_mean_ms"],
                      "device_kernels_ms_per_batch": r["kernel_s"] * 1e3, "reviews_beyond_limits": r["too_big"]}
            if oracle is not None and first:
                n = min(oracle_n, stream_args.batch) // 64 * 64
                stream["parity_python_oracle"] = python_oracle_leg(templates, constraints, batch, first[0], n, ids=ids, oracle=oracle)
                stream["parity_python_oracle"]["geometry"] = "the first streamed batch (objects [0, %d) of the same stream): a %d-review table built, evaluated and downloaded by the streaming pipeline" % (stream_args.batch, stream_args.batch)
            if oracle_n > 0 and first and stream_args.batch <= reviews:
                try:   # every object of that streamed batch against the independent compiled checker
                    _, stream["parity_compiled_independent"] = indep_leg(templates, constraints, batch, first[0], ids=ids, budget_s=0.2, n=stream_args.batch)
                    stream["parity_compiled_independent"].pop("_results_by_row", None)
                except Exception as ex:   # noqa: BLE001
                    stream["parity_compiled_independent"] = {"error": "%s: %s" % (type(ex).__name__, ex)}
        except Exception as ex:   # noqa: BLE001
            stream = {"error": "%s: %s" % (type(ex).__name__, ex)}
    batch.free()
    drv.engine.close()
    return out, stream


def other_configs(args, dev_index, dev, fx, nss, budget_s=200.0, known_prefix=None):
    """configs[1], configs[4] (resident + streaming) and the N = 1 point of configs[3], each in its own engine; a leg is skipped
    (and says so) once the budget is spent so that the default run stays within a few minutes."""
    t_start = time.perf_counter()
    detail, brief = {}, {}

    def left():
        return budget_s - (time.perf_counter() - t_start)

    def parity_brief(pp):
        if not pp:
            return None
        if "error" in pp:
            return {"error": pp["error"][:80]}
        return {"n": pp["n"], "equal": pp["pairs_equal"], "pairs": pp["oracle_violating_pairs"], "s": _sig(pp["seconds"], 3)}

    def run(name, fn):
        if left() <= 0:
            detail[name] = brief[name] = {"skipped": "time budget of the default run spent"}
            return None
        try:
            t = time.perf_counter()
            r = fn()
            r[0]["leg_seconds"] = time.perf_counter() - t
            return r
        except Exception as ex:   # noqa: BLE001
            detail[name] = brief[name] = {"error": ("%s: %s" % (type(ex).__name__, ex))[:200]}
            return None

    def resident_brief(d):
        rf = d["roofline"]
        return {"w": "%dx%d" % (d["constraints"], d["reviews"]), "ms": _sig(d["ms_per_step"]), "evals_s": _sig(d["evals_per_s"]), "frac": _sig(rf["frac"], 3),
                "GBs": _sig(rf["achieved"]), "groups": d["plan_groups"], "parity": parity_brief(d.get("parity_python_oracle")),
                "compiled": {k: (_sig(v, 3) if isinstance(v, float) else v) for k, v in (d.get("parity_compiled_independent") or {}).items() if k in ("n", "pairs_equal", "seconds", "error")},
                "messages": {k: v for k, v in (d.get("parity_messages_compiled_independent") or {}).items() if k in ("objects", "messages", "messages_equal", "error")},
                "leg_s": _sig(d["leg_seconds"], 3), "first_sweep_s": _sig(d.get("first_sweep_s"), 3), "first_sweep_warm_s": _sig(d.get("first_sweep_warm_s"), 3)}
    r = run("configs1", lambda: side_point(1, 100000, max(args.steps, 50), args.warmup, args.side_oracle_sample, dev_index, fx, nss))
    if r:
        detail["configs1"], brief["configs1"] = r[0], resident_brief(r[0])
    import copy
    sa = copy.copy(args)
    # (batches of 16 384 reviews: at the offered 10^6/s a batch of 65 536 answered in 8.3 ms mean / 9.8-12 ms p99, one of 16 384 in 3.8 / 5.6 ms,
    #  and the closed loop moves 5.7 M reviews/s instead of 4.1 M -- profiles/r05_stream_y_batch_sizes.log;
    #  eight warm-up batches: pool growth and the first uploads of a new batch geometry showed as one 15 ms batch among the first timed ones)
    sa.stream_batches, sa.warmup, sa.stream_unique, sa.batch, sa.offered = 64, 8, 8, 16384, 1e6
    r = run("configs4", lambda: side_point(4, 200000, max(args.steps, 20), args.warmup, args.side_oracle_sample, dev_index, fx, nss, with_stream=True, stream_args=sa, dev=dev, totals=True, warm_probe=True))
    if r:
        detail["configs4"], brief["configs4"] = r[0], resident_brief(r[0])
        tl = r[0].get("audit_result_totals") or {}
        brief["configs4"]["totals"] = ({"error": tl["error"][:80]} if "error" in tl else
                                       {"s": _sig(tl.get("seconds"), 3), "first_s": _sig(tl.get("seconds_first_call"), 3), "rendered_share": _sig(tl.get("rendered_share"), 3),
                                        "equal": (tl.get("host_pass_over_every_pair") or {}).get("equal"), "host_pass_s": _sig((tl.get("host_pass_over_every_pair") or {}).get("seconds"), 3),
                                        "equal_compiled_checker": (tl.get("independent_compiled_checker") or {}).get("equal")})
        detail["configs4_stream"] = r[1]
        if r[1] and "error" not in r[1]:
            brief["configs4_stream"] = {"offered": r[1]["offered_reviews_per_s"], "achieved": _sig(r[1]["achieved_reviews_per_s"]), "p50_ms": _sig(r[1]["batch_latency_ms"]["p50"], 3),
                                        "p99_ms": _sig(r[1]["batch_latency_ms"]["p99"], 3), "parity": parity_brief(r[1].get("parity_python_oracle")),
                                        "compiled": {k: (_sig(v, 3) if isinstance(v, float) else v) for k, v in (r[1].get("parity_compiled_independent") or {}).items() if k in ("n", "pairs_equal", "seconds", "error")}}
        else:
            brief["configs4_stream"] = r[1]
    r = run("configs3_n1", lambda: side_point(2, 10000000, max(args.steps, 20), 3, 0, dev_index, fx, nss, strided_n=args.strided_sample, known_prefix=known_prefix))
    if r:
        d = r[0]
        d["note"] = "the origin of configs[3]'s strong-scaling curve: all 10 M objects on ONE MI355X (N > 1: bench.py --gpus N --scaling strong --reviews 10000000); the table is far beyond the 256 MiB Infinity Cache"
        detail["configs3_n1"] = d
        brief["configs3_n1"] = dict(resident_brief(d), ingest_s=_sig(d["ingest"]["flatten_s"] + d["ingest"]["h2d_s"], 3), table_GB=_sig(d["table_bytes"] / 1e9, 3))
        pc = d.get("parity_compiled_independent") or {}
        # (`parity`: the leg this table has -- the compiled checker over the strided sample, incl. the first, middle and last row groups)
        brief["configs3_n1"]["parity"] = ({"error": pc["error"][:80]} if "error" in pc else
                                          {"n": pc.get("n"), "equal": pc.get("pairs_equal"), "totals_equal": pc.get("per_constraint_totals_equal"), "pairs": pc.get("checker_violating_pairs"),
                                           "last_word": pc.get("last_word"), "table_words": pc.get("table_words"), "prefix_1M_equal": (pc.get("prefix_totals") or {}).get("equal"), "s": _sig(pc.get("seconds"), 3)}) if pc else None
    return detail, brief


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=200)
    ap.add_argument("--warmup", type=int, default=10)
    ap.add_argument("--config", type=int, default=None, choices=[1, 2, 4],
                    help="1: 30 PSP x Pods (configs[1]); 2: 50 constraints x mixed objects (configs[2]); 4: the 200-template policy corpus x mixed objects (configs[4]'s policy set)")
    ap.add_argument("--reviews", type=int, default=None, help="objects per GPU (weak) / in total (strong); default 1000000 (config 2), 100000 (config 1)")
    ap.add_argument("--scaling", choices=["weak", "strong"], default="weak")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--lean", action="store_true", help="tuning runs: the timed sweep and the roofline figures only (no RESULT totals, no CPU / oracle legs)")
    ap.add_argument("--streaming", action="store_true", help="configs[4] as a STREAM: batches of --batch reviews through ingest -> H2D -> launch -> D2H, "
                    "double-buffered, round robin over the ranks; reports the achieved rate and the batch latency percentiles (steps = --stream-batches)")
    ap.add_argument("--batch", type=int, default=16384, help="reviews per streamed batch")
    ap.add_argument("--stream-batches", type=int, default=64, help="timed batches over all ranks")
    ap.add_argument("--stream-unique", type=int, default=8, help="distinct pre-generated batches per rank (cycled)")
    ap.add_argument("--offered", type=float, default=1e6, help="offered load in reviews/s over all ranks (0: closed loop, as fast as the pipeline goes)")
    ap.add_argument("--oracle-sample", type=int, default=1048576, help="objects of the timed table the pure-Python oracle re-evaluates (parity_python_oracle)")
    ap.add_argument("--side-oracle-sample", type=int, default=16384, help="... of every other_configs table")
    ap.add_argument("--no-prune", action="store_true", help="tables with a row for every key path of every object (rounds 1-3) instead of GK_TABLE_PRUNED: rows of the key "
                    "paths the loaded constraints read; the kernel's algorithmic bytes are the same, the ingest and the table are not")
    ap.add_argument("--no-other-configs", action="store_true", help="only the headline workload (the default run adds configs[1], configs[4] resident + streaming "
                    "and the N = 1 point of configs[3] as `other_configs`, each with its own parity leg)")
    ap.add_argument("--strided-sample", type=int, default=1 << 20, help="objects of the 10 M-object configs[3] table the independent compiled checker re-evaluates "
                    "(whole bitmap words: the first, middle and last 2048 words + an even spread)")
    ap.add_argument("--test-hostemu", action="store_true", help="TEST ONLY (tests/test_bench_dist.py): the CPU emulation build of the engine and gloo instead of an MI355X "
                    "and RCCL, to exercise this file's sharded path in the GPU-less build container; the line says so (`emulated`) and is not a measurement")
    args = ap.parse_args()
    # ---- ranks.  The driver starts N > 1 as `python -m torch.distributed.run --nproc-per-node N bench.py --gpus N` (WORLD_SIZE set);
    # a bare `python bench.py --gpus N` starts its own N ranks the same way.  Either way the line is printed only if the job really
    # has N ranks: a world size that differs from --gpus is an error (exit 2), never a silent N = 1 line.
    if "WORLD_SIZE" not in os.environ and args.gpus > 1:
        import subprocess
        if not args.test_hostemu:
            import torch
            have = torch.cuda.device_count() if torch.cuda.is_available() else 0
            if have < args.gpus:
                sys.stderr.write("bench.py --gpus %d: only %d GPU(s) visible on this box -- refusing to print a line for fewer ranks than asked for\n" % (args.gpus, have))
                sys.exit(2)
        import socket
        with socket.socket() as sk:
            sk.bind(("127.0.0.1", 0))
            port = sk.getsockname()[1]
        cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(args.gpus), "--master-addr", "127.0.0.1", "--master-port", str(port),
               os.path.abspath(__file__)] + sys.argv[1:]
        sys.exit(subprocess.call(cmd))
    env_world = int(os.environ.get("WORLD_SIZE", "1"))
    if env_world != args.gpus:
        sys.stderr.write("bench.py: --gpus %d but the launcher started %d rank(s) (WORLD_SIZE): refusing to print a line for another job size\n" % (args.gpus, env_world))
        sys.exit(2)
    want_others = args.config is None and not args.no_other_configs and not args.lean and not args.streaming and args.scaling == "weak" and args.reviews is None
    if args.config is None:
        args.config = 2
    if args.reviews is None:
        args.reviews = {1: 100000, 2: 1000000, 4: 200000}[args.config]

    import torch
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    dist = None
    emu = bool(args.test_hostemu)
    if world > 1 or os.environ.get("GK_FORCE_DIST"):   # GK_FORCE_DIST=1: exercise the sharded path on one GPU (world size 1)
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29531")
        os.environ.setdefault("RANK", "0")
        os.environ.setdefault("WORLD_SIZE", "1")
        if emu:
            dist.init_process_group("gloo")
        else:
            torch.cuda.set_device(local_rank)
            dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))
        if dist.get_world_size() != args.gpus:
            sys.stderr.write("bench.py: --gpus %d but the process group has %d rank(s)\n" % (args.gpus, dist.get_world_size()))
            sys.exit(2)
    if emu:
        dev = torch.device("cpu")
        torch.cuda.synchronize = lambda *a, **k: None   # (this process only: the emulation has no device to wait for)
    else:
        assert torch.cuda.is_available(), "bench.py needs an MI355X (there is no CPU fallback)"
        dev = torch.device("cuda", local_rank)
        torch.cuda.set_device(dev)

    from gatekeeper_amd import driver as D
    from gatekeeper_amd import synth
    from gatekeeper_amd.sweep import ShardedSweep

    fx = synth.load_fixtures()
    templates = synth.psp_templates(fx)
    constraints = synth.psp_constraints() if args.config == 1 else synth.audit_constraints()
    if args.config == 4:
        templates, constraints = synth.corpus(fx)
    nss = synth.gen_namespaces()
    if args.scaling == "strong":
        per_rank = (args.reviews + world - 1) // world
        start, n_local = rank * per_rank, max(0, min(per_rank, args.reviews - rank * per_rank))
        total_reviews = args.reviews
    else:
        start, n_local = rank * args.reviews, args.reviews
        total_reviews = args.reviews * world

    drv = D.Driver(device=local_rank, hostemu=emu)
    client = D.Client(drv)
    for t in templates:
        client.AddTemplate(t)
    for k in constraints:
        client.AddConstraint(k)
    defaulted = [client.constraints[(k["kind"], k["metadata"]["name"])] for k in constraints]
    batch_constraint_ids[:] = [drv.constraint_id(c) for c in defaulted]

    if args.streaming:
        if args.warmup > 8:
            args.warmup = 2 * world      # (the non-streaming default of 10 sweeps would be 10 batches of warm-up)
        r = stream_leg(args, drv, client, templates, constraints, nss, rank, world, dev, dist)
        if rank == 0:
            nc = len(constraints)
            evals = float(nc) * r["reviews_all"]
            achieved = r["algo_bytes"] / r["kernel_s"] / 1e9 if r["kernel_s"] > 0 else 0.0
            out = {"metric": "AdmissionReview x constraint evals/sec", "value": evals / r["dt"], "unit": "evals/s", "n_gpus": world,
                   "steps": args.stream_batches, "warmup": args.warmup, "ms_per_step": r["dt"] / max(1, r["batches_rank0"]) * 1e3, "higher_is_better": True,
                   "scaling": "weak", "vs_baseline": None, "dtype": "u32", "data": "synthetic",
                   "config": {"workload": "configs[4]: %d ConstraintTemplates + constraints (required labels/allowedRegex, allowed repos, banned image tags, container "
                                          "limits, required probes, PSP x 5; namespace globs), STREAMING: %d mixed synthetic reviews as JSON text in batches of %d, "
                                          "offered %.0f reviews/s, batches round robin over %d GPU(s), double-buffered ingest -> H2D -> launch -> D2H per batch"
                                          % (nc, r["reviews_all"], args.batch, args.offered, world),
                              "constraints": nc, "reviews_total": r["reviews_all"], "batch": args.batch, "offered_reviews_per_s": args.offered,
                              "timed_region_s": r["dt"], "parallelism": "batches round robin over %d GPU(s), no collective" % world,
                              "violating_pairs_per_batch": r["pairs_per_batch"], "reviews_beyond_limits": r["too_big"], "unique_batches_per_rank": r["unique_batches"]},
                   "stream": {"achieved_reviews_per_s": r["reviews_all"] / r["dt"], "batch_latency_ms": r["lat_ms"], "stage_mean_ms": r["stage_mean_ms"],
                              "what": "latency = bitmaps on the host - arrival of the batch (open loop at the offered rate; queueing included); "
                                      "value counts host ingest + PCIe: this is the end-to-end streaming rate, host-bound"},
                   "roofline": {"bound": "hbm", "kernel": "gk_jit_tiles", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": achieved / HBM_PEAK_GBS,
                                "traffic": None, "algo_bytes_per_launch": r["algo_bytes"], "avg_kernel_ms": r["kernel_s"] * 1e3, "lds_bytes_per_tile": r["lds"],
                                "note": "dominant kernel per plan group launch over one %d-review batch (launch-bound at this size); the resident-sweep record is the roofline line" % args.batch}}
            print(json.dumps(out))
        if dist is not None:
            dist.destroy_process_group()
        return

    # objects [start, start + n_local) of the global synthetic stream, as JSON text (native generator == synth.py)
    t_gen = time.perf_counter()
    batch = synth.NativeBatch(drv.engine.lib, n_local, seed=synth.SEED, mixed=(args.config != 1), start=start, namespaces=nss)
    t_gen = time.perf_counter() - t_gen
    # end-to-end leg: JSON -> parse -> HandleReview -> flatten -> HBM (this is what a non-resident review costs)
    table = drv.engine.create_table_native(batch.reviews, n_local, keep_docs=False, resident=True, keep_text=True, pruned=not getattr(args, "no_prune", False))   # (text kept by `batch`: RESULT totals below)
    st = table.stats()
    st_again = None
    if not args.lean and n_local <= 2_000_000:   # the same batch once more: what ingest costs once the host threads' caches, the staging pool and the path tables are warm
        again = drv.engine.create_table_native(batch.reviews, n_local, keep_docs=False, resident=True, pruned=not getattr(args, "no_prune", False))
        st_again = again.stats()
        again.free()
    sweep = ShardedSweep(client, table=table, n=n_local, dist=dist, device=dev)
    # the job size as the engine's communicator itself reports it (ncclCommCount): must be the --gpus asked for on EVERY rank
    rccl_rank, rccl_ranks = sweep.comm_info() if dist is not None else (0, None)
    if dist is not None and (rccl_ranks != args.gpus or rccl_rank != rank):
        sys.stderr.write("bench.py rank %d: the engine's communicator reports rank %d of %d, --gpus is %d\n" % (rank, rccl_rank, rccl_ranks, args.gpus))
        sys.exit(2)

    def barrier():
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize()

    def local(steps, download=False, time_each=False, kernel_only=False):
        """plain launches of the hot path over this rank's shard, no exchange"""
        for _ in range(steps):
            table.launch(time_each=time_each, kernel_only=kernel_only)
        return table.eval(download=download, collect_only=True)

    def sharded_step(steps):
        return sweep.sweep(steps, collect=True)           # `steps` enqueue-only passes, then the answer of the last

    step = sharded_step if dist is not None else local    # sharded: local evaluation + the engine's RCCL exchange per step
    if args.warmup:
        step(args.warmup)
    barrier()
    t0 = time.perf_counter()
    step(args.steps)                       # exactly `steps` passes (+ exchanges when sharded) in the timed region
    barrier()
    dt = time.perf_counter() - t0
    sharded = sweep.sweep(1, download=False) if dist is not None else None
    res = local(min(args.steps, 20))       # launch-to-launch average of the whole step (and the per-launch byte accounting)
    # The dominant kernel's own duration by HIP events on its stream, two ways: (a) ONE event pair around `steps` consecutive launches
    # of that kernel alone (GK_EVAL_KERNEL_ONLY: no totals kernel in between) -- the average includes the gap between two launches;
    # (b) an event pair per launch (GK_EVAL_TIME_EACH) -- every bracket includes the dispatch of the kernel behind an event record,
    # 3-4 us on a 65 us kernel (round 5: 68.2 us by (b), 64.3 us by rocprofv3's kernel trace of the same build).  `roofline` uses (a).
    b2b = local(min(args.steps, 20), kernel_only=True) if dist is None else None
    final = local(1, download=True)
    counts = final.counts
    # isolated kernel duration: a second, untimed pass with one HIP event pair per launch (the timed region above
    # brackets all launches with one pair, i.e. its average includes the gaps between consecutive launches)
    iso = local(min(args.steps, 20), time_each=True)
    if dist is not None:
        tmax = torch.tensor([dt], device=dev, dtype=torch.float64)
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        dt = float(tmax.item())

    if rank == 0:
        nc = len(constraints)
        evals = float(nc) * total_reviews * args.steps
        # average duration of the dominant kernel alone: back to back under one event pair when there is one plan group, else per-launch events
        kernel_b2b_ms = float(b2b.fast_kernel_ms) if (b2b is not None and len(constraints) <= 64 and b2b.fast_kernel_ms > 0) else None
        kernel_s = (kernel_b2b_ms if kernel_b2b_ms is not None else iso.fast_kernel_ms) / 1e3
        achieved = res.algo_bytes / kernel_s / 1e9 if kernel_s > 0 else 0.0
        full_table_bytes = int(res.n_rows) * 16 + n_local * 4   # what a kernel streaming every row would read
        cfg_name = ("configs[1]: 30 gatekeeper PSP constraints (5 in-tree PSP templates x 6 parameterisations) x %d synthetic Pod "
                    "AdmissionReviews" if args.config == 1 else
                    "configs[4] policy set: 200 ConstraintTemplates + 200 constraints (required labels/allowedRegex, allowed repos, banned image tags, "
                    "container limits, required probes, PSP x 5; namespace globs) x %d mixed synthetic cluster objects" if args.config == 4 else
                    "configs[2]: pkg/audit sweep, 50 constraints x %d mixed synthetic cluster objects (80%% Pod, 10%% Deployment, 5%% Namespace, "
                    "5%% Service/ConfigMap)") % total_reviews
        e2e_s = st["flatten_s"] + st["upload_s"]
        out = {
            "metric": "AdmissionReview x constraint evals/sec",
            "value": evals / dt, "unit": "evals/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": dt / args.steps * 1e3, "higher_is_better": True, "scaling": args.scaling, "vs_baseline": None,
            "dtype": "u32", "data": "synthetic" if not emu else "synthetic -- CPU EMULATION of the kernels (--test-hostemu): plumbing test, not a measurement",
            # ranks of the engine's own communicator, read back from RCCL (ncclCommCount) after gk_comm_init; null: one process, no exchange step
            "rccl_ranks": rccl_ranks,
            "config": {"workload": cfg_name, "constraints": nc, "reviews_total": total_reviews, "reviews_rank0": n_local,
                       "rows_rank0": int(res.n_rows), "rows_read_rank0": int(res.n_rows_read), "table_bytes_rank0": int(st["device_bytes"]),
                       "timed_region_s": dt,
                       "parallelism": ("objects block-sharded across %d GPUs; per sweep ONE in-place ncclAllGather of [violation bitmaps | counts | fail-closed counts] "
                                       "issued by the engine on the kernel's stream (global totals = sums over the gathered slot tails); four enqueues "
                                       "per pass, no host round trip" % world) if dist is not None else "1 GPU",
                       "global_violating_pairs": int(sharded.totals.sum()) if sharded is not None else int(counts.sum()),
                       "violating_pairs_rank0": int(counts.sum()), "reviews_beyond_limits_rank0": len(final.too_big_reviews()),
                       # reviews beyond the DEVICE's limits that the engine's exact host evaluator answered instead of refusing them
                       "reviews_evaluated_on_host_rank0": len(getattr(final, "host_evaluated", []))},
            # N > 1: what the exchange step costs next to the local sweep (rank 0's figures of one collecting sweep behind the timed region:
            # HIP events around the all-gather alone; the timed passes are enqueue-only and carry no events).  `ms_per_step` against
            # sweep_ms_local + exchange_ms says whether the overlapped exchange (GK_SHARD_OVERLAP=1, opt-in) hid it.
            "exchange": None if sharded is None else {
                "exchange_bytes_per_rank": sharded.exchange_bytes_inbound, "exchange_ms": sharded.exchange_ms, "sweep_ms_local": float(sharded.fast_kernel_ms),
                "overlap_enabled": sharded.exchange_overlapped, "slot_bytes": int(sharded.slot_bytes),
                "exchange_GBps_inbound": sharded.exchange_bytes_inbound / (sharded.exchange_ms * 1e6) if sharded.exchange_ms > 0 else None,
                "hidden_by_overlap": (dt / args.steps * 1e3) < 0.75 * (float(sharded.fast_kernel_ms) + sharded.exchange_ms) if sharded.exchange_ms > 0 else None,
                "bound": ("exchange" if sharded.exchange_ms > float(sharded.fast_kernel_ms) else "sweep") if sharded.exchange_ms > 0 else None},
            # (the plan-specialised build -- what rocprofv3 shows for this workload; GK_NO_JIT=1 runs the generic bytecode build instead)
            "roofline": {"bound": "hbm", "kernel": "gk_eval_tiles_256" if os.environ.get("GK_NO_JIT") else "gk_jit_tiles", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": achieved / HBM_PEAK_GBS, "traffic": None, "algo_bytes_per_launch": int(res.algo_bytes),
                         "avg_kernel_ms": kernel_s * 1e3, "launches_timed": int((b2b if kernel_b2b_ms is not None else iso).n_launches),
                         "clock": ("HIP events on the launch stream: one pair around %d consecutive launches of the kernel alone (GK_EVAL_KERNEL_ONLY)" % int(b2b.n_launches)) if kernel_b2b_ms is not None
                                  else "HIP events on the launch stream: one pair per launch (GK_EVAL_TIME_EACH)",
                         "avg_kernel_ms_event_pair_per_launch": iso.fast_kernel_ms,
                         # (one event pair around all launches of a view; the plan groups of a >64-formula constraint set share
                         #  the stream, so the figure is only meaningful for a single group)
                         "avg_launch_ms_back_to_back": res.fast_kernel_ms if nc <= 64 else None, "lds_bytes_per_tile": int(res.lds_bytes), "kernel_text_hash": "%016x" % int(res.kernel_text_hash),
                         "full_table_bytes": full_table_bytes,
                         "full_table_GBps": full_table_bytes / kernel_s / 1e9 if kernel_s > 0 else None,
                         "kernel_only_evals_per_s": nc * n_local / kernel_s if kernel_s > 0 else None},
            "end_to_end": {"what": "rank 0: JSON text -> parse -> HandleReview -> flatten -> row groups -> HBM for the %d objects of its shard "
                                   "(gk_table_create), then one sweep" % n_local,
                           "flatten_s": st["flatten_s"], "h2d_s": st["upload_s"], "sweep_s": dt / args.steps, "host_threads": st["host_threads"],
                           "host_cpus_usable": int(drv.engine.lib.gk_host_cpus()), "host_hardware_threads": os.cpu_count(),
                           "json_bytes": st["json_bytes"], "reviews_per_s": n_local / e2e_s if e2e_s > 0 else None,
                           "evals_per_s": nc * n_local / (e2e_s + dt / args.steps) if e2e_s > 0 else None,
                           "json_MBps": st["json_bytes"] / st["flatten_s"] / 1e6 if st["flatten_s"] > 0 else None,
                           "second_table_of_the_same_batch": None if not st_again else {"flatten_s": st_again["flatten_s"], "h2d_s": st_again["upload_s"],
                                                                                        "json_MBps": st_again["json_bytes"] / st_again["flatten_s"] / 1e6 if st_again["flatten_s"] > 0 else None},
                           "generate_s": t_gen},
        }
        if args.config == 2 and n_local == 1000000 and world == 1 and kernel_s > 0 and not os.environ.get("GK_DICT_MATCH"):
            # Round 5 moved the match layer's five string facts per review (5 rows + 5 string headers = 160 of ~380 bytes) into ONE
            # dictionary row: the sweep streams 234 MB where rounds 1-4 streamed 379 MB -- `frac` is priced on what
            # is streamed NOW, so a launch that got a third faster shows a LOWER fraction.  For the comparison with earlier rounds:
            out["roofline"]["round4_layout"] = {"algo_bytes_per_launch": 378794260, "frac_of_this_launch_time_at_those_bytes": 378794260 / kernel_s / 1e9 / HBM_PEAK_GBS,
                                                "what": "the same launch time priced at the bytes the round-4 table layout streamed for this workload (GK_DICT_MATCH=0 builds that layout: 0.1136 ms = 0.417 on the box that ran this layout in 0.0763 ms, profiles/r05_variants_r_match_dictionary.log)"}
        if emu:
            out["emulated"] = True
        # HBM traffic of the dominant kernel from the committed rocprofv3 PMC passes of this same command (bench.py
        # cannot run a profiler around itself); only reported when the profiled workload is the one just timed
        try:
            pmc = json.load(open(os.path.join(ROOT, "profiles", "pmc_latest.json")))
            # (... and the profiled build streamed the bytes this one does: a PMC figure of an earlier table layout says nothing)
            same_bytes = pmc.get("algo_bytes_per_launch") is None or abs(pmc["algo_bytes_per_launch"] - int(res.algo_bytes)) <= 0.02 * int(res.algo_bytes)
            # (... and ran the kernel text this run timed: the passes are stamped with the hash bench.py printed under them -- a PMC
            #  figure measured on another kernel or table layout can never ride this line)
            same_kernel = pmc.get("kernel_text_hash") == "%016x" % int(res.kernel_text_hash)
            if pmc.get("config") == args.config and pmc.get("reviews") == n_local and world == 1:
                if same_bytes and same_kernel:
                    out["roofline"]["traffic"] = pmc["hbm_bytes_per_launch"]
                    out["roofline"]["traffic_source"] = pmc["source"]
                    # The SECOND roof.  Round 6 cut the bytes a sweep streams by 45 % (element carriers, the review facts row) and the
                    # kernel's time by 17 %: `frac` -- bytes over time against the HBM peak -- FELL, although nothing about the memory
                    # system got worse.  What binds the kernel is instruction issue at the occupancy its LDS footprint allows; the
                    # SQ counters of the same PMC passes say how close to THAT roof it runs: wave64 VALU instructions per launch
                    # against what the device's SIMDs can issue in the kernel's duration (a wave64 VALU instruction occupies a
                    # SIMD-32 for two cycles: 256 CUs x 4 SIMDs x 2.4 GHz / 2), and the share of wave cycles with an instruction in flight.
                    sq = pmc.get("sq") or {}
                    if sq.get("SQ_INSTS_VALU") and kernel_s > 0:
                        peak_valu = 256 * 4 * 2.4e9 / 2.0
                        out["roofline"]["issue"] = {
                            "valu_wave_insts_per_launch": sq.get("SQ_INSTS_VALU"), "salu_wave_insts_per_launch": sq.get("SQ_INSTS_SALU"), "lds_wave_insts_per_launch": sq.get("SQ_INSTS_LDS"),
                            "valu_issue_frac": sq["SQ_INSTS_VALU"] / kernel_s / peak_valu, "peak_valu_wave_insts_per_s": peak_valu,
                            "waves_waiting_share": (sq.get("SQ_WAIT_ANY") / sq["SQ_WAVE_CYCLES"]) if sq.get("SQ_WAVE_CYCLES") else None,
                            "waves_issuing_share": (sq.get("SQ_ACTIVE_INST_ANY") / sq["SQ_WAVE_CYCLES"]) if sq.get("SQ_WAVE_CYCLES") else None,
                            "what": "rocprofv3 --pmc SQ_* of the same command and kernel text (profiles/pmc_latest.json); valu_issue_frac = VALU wave instructions / (kernel time x 256 CUs x 4 SIMD-32 x 2.4 GHz / 2 cycles per wave64 instruction)"}
                else:
                    out["roofline"]["traffic_dropped"] = ("profiles/pmc_latest.json was measured on kernel text %s streaming %s bytes per launch; this run timed kernel text %016x "
                                                          "streaming %d: the stale figure is not reported" % (pmc.get("kernel_text_hash"), pmc.get("algo_bytes_per_launch"),
                                                                                                             int(res.kernel_text_hash), int(res.algo_bytes)))
        except (OSError, ValueError):
            pass
        # the audit's RESULT totals (pkg/audit/manager.go:893-904: totalViolationsPerConstraint counts types.Results, several per
        # violating pair) of the MEASURED table.  Round 3: the DEVICE decides, per violating pair, whether the template can yield
        # more than one result for the review (the totals plans: Template::compile_multi); a pair it does not flag counts one
        # result unrendered, only the flagged ones are parsed from the batch's JSON text and rendered on the host workers.
        # Checked here against the host pass over EVERY violating pair (GK_TOTALS_RENDER_ALL=1, the round-2 path).
        try:
            if args.lean:
                raise RuntimeError("skipped (--lean)")
            out["audit_result_totals"] = totals_leg(table)
        except Exception as ex:   # noqa: BLE001
            out["audit_result_totals"] = {"error": "%s: %s" % (type(ex).__name__, ex)}
        if not args.no_cpu_baseline and not args.lean and world == 1:   # (the checkers and the CPU baseline: N = 1 only -- at N > 1 the other ranks would wait ~2 min for rank 0)
            # cpu_baseline AND the parity checker are the INDEPENDENT compiled restatement (oracle/libgkindep.so: nothing of the product
            # linked).  Until round 4 `parity_sample` came from oracle/cpu_ref.cpp, whose JSON reader and Rego evaluator are the product's
            # own host objects -- product against product for the Rego half; that loop is no part of the bench line any more.
            try:
                out["cpu_baseline"], out["parity_compiled_independent"] = indep_leg(templates, constraints, batch, final)
            except Exception as ex:   # noqa: BLE001
                out["cpu_baseline"] = {"error": "%s: %s" % (type(ex).__name__, ex)}
                out["parity_compiled_independent"] = {"error": "%s: %s" % (type(ex).__name__, ex)}
            out["parity_sample"] = {k: v for k, v in out["parity_compiled_independent"].items() if k in ("n", "constraints", "pairs_equal", "device_violating_pairs", "checker_violating_pairs", "checker", "error")}
            try:   # ... the RESULT totals the checker counted in that pass against gk_table_totals
                vs = totals_against_checker(table, out["parity_compiled_independent"])
                if vs is not None and isinstance(out.get("audit_result_totals"), dict):
                    out["audit_result_totals"]["independent_compiled_checker"] = vs
            except Exception as ex:   # noqa: BLE001
                out["audit_result_totals"] = dict(out.get("audit_result_totals") or {}, independent_compiled_checker={"error": "%s: %s" % (type(ex).__name__, ex)})
            out["parity_compiled_independent"].pop("_results_by_row", None)
            try:   # ... and the TEXT of the messages, on a prefix of the table
                out["parity_messages_compiled_independent"] = messages_leg(templates, constraints, batch, final, table)
            except Exception as ex:   # noqa: BLE001
                out["parity_messages_compiled_independent"] = {"error": "%s: %s" % (type(ex).__name__, ex)}
            try:
                out["parity_python_oracle"] = python_oracle_leg(templates, constraints, batch, final, args.oracle_sample)
            except Exception as ex:   # the checker must not cost the bench line; an absent leg is visible as such
                out["parity_python_oracle"] = {"error": "%s: %s" % (type(ex).__name__, ex)}
        # every other BASELINE config, each with its own independent parity leg at the bench's geometry.  LAST in the line and
        # compact: the driver's record keeps the tail of the line; the full records sit in other_configs_detail before it.
        if want_others and world == 1:
            sweep = table = batch = None   # (the headline table and its text leave HBM / host memory before the 10 M-object point)
            import gc
            gc.collect()
            row_of0 = {int(cid): i for i, cid in enumerate(final.constraint_ids)}
            known = (n_local, [int(counts[row_of0[cid]]) for cid in batch_constraint_ids]) if n_local % 64 == 0 else None
            detail, brief = other_configs(args, local_rank, dev, fx, nss, known_prefix=known)
            out["other_configs_detail"] = detail
            pp = out.get("parity_python_oracle") or {}
            brief["configs2"] = {"w": "%dx%d" % (nc, total_reviews), "ms": _sig(out["ms_per_step"]), "frac": _sig(out["roofline"]["frac"], 3),
                                 "parity": {"n": pp.get("n"), "equal": pp.get("pairs_equal"), "pairs": pp.get("oracle_violating_pairs"), "s": _sig(pp.get("seconds"), 3)},
                                 "compiled_independent_parity": {k: (out.get("parity_compiled_independent") or {}).get(k) for k in ("n", "pairs_equal", "checker_violating_pairs", "seconds", "error") if k in (out.get("parity_compiled_independent") or {})},
                                 "messages": {k: v for k, v in (out.get("parity_messages_compiled_independent") or {}).items() if k in ("objects", "messages", "messages_equal", "error")},
                                 "totals": {"s": _sig((out.get("audit_result_totals") or {}).get("seconds"), 3), "rendered_share": _sig((out.get("audit_result_totals") or {}).get("rendered_share"), 3),
                                            "equal": ((out.get("audit_result_totals") or {}).get("host_pass_over_every_pair") or {}).get("equal"),
                                            "equal_compiled_checker": ((out.get("audit_result_totals") or {}).get("independent_compiled_checker") or {}).get("equal")}}
            e2e = out.get("end_to_end") or {}
            again = e2e.get("second_table_of_the_same_batch") or {}
            brief["configs2"]["ingest"] = {"first_table_json_MBps": _sig(e2e.get("json_MBps"), 4), "first_table_flatten_s": _sig(e2e.get("flatten_s"), 3),
                                           "second_table_json_MBps": _sig(again.get("json_MBps"), 4), "second_table_flatten_s": _sig(again.get("flatten_s"), 3),
                                           "host_threads": e2e.get("host_threads"), "host_cpus_usable": e2e.get("host_cpus_usable")}
            # `roofline` beside the table no cache can hold: configs[3]'s N = 1 point (10 M objects, > 1 GB streamed per sweep, far beyond
            # the 256 MiB Infinity Cache -- configs[2]'s 130 MB fit in it, and FETCH_SIZE counts cache hits)
            d3 = detail.get("configs3_n1") or {}
            if "roofline" in d3:
                r3 = d3["roofline"]
                out["roofline"]["uncached_table"] = {"workload": d3.get("workload"), "algo_bytes_per_launch": r3.get("algo_bytes_per_sweep_table_once"), "avg_kernel_ms": r3.get("seconds", 0) * 1e3,
                                                     "achieved": r3.get("achieved"), "frac": r3.get("frac"), "ms_per_step": d3.get("ms_per_step"), "table_bytes": d3.get("table_bytes")}
            out["other_configs"] = brief
        print(json.dumps(out))
    if dist is not None:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
