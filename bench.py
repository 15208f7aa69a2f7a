#!/usr/bin/env python3
"""Benchmark of the hot path: AdmissionReview x constraint evaluations per second (BASELINE.json `metric`).

A "step" = one pass of the hot path (Match + violation predicates for every loaded constraint) over one batch of
synthetic reviews that is already resident in HBM.  Default workload = BASELINE.json configs[1]:
30 PSP constraints x 100k synthetic Pod reviews on one MI355X.  With --gpus N (launched by torch.distributed.run,
one process per GPU) every rank sweeps its own shard of N x 100k objects (weak scaling) and the per-shard violation
bitmaps / counts are exchanged with RCCL all-gather / all-reduce inside every step.

Prints ONE JSON line on rank 0 (contract in the task description) with `roofline` and `cpu_baseline` objects.
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0   # MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec


def cpu_baseline(templates, constraints, objs, nss, budget_s=15.0):
    """The oracle (Python restatement of the reference's serial audit loop, pkg/audit/manager.go:591-642: per object
    Client.Review = per-constraint match + Rego evaluation) timed on ONE host core over a bounded sample."""
    from oracle import client as OC
    from oracle import target as OT
    from gatekeeper_amd import synth
    oc = OC.Client()
    for t in templates:
        oc.add_template(t)
    for k in constraints:
        oc.add_constraint(k)
    t0 = time.perf_counter()
    n = 0
    for o in objs:
        oc.review(OT.AugmentedUnstructured(OT.Unstructured(o), synth.namespace_for(o, nss), "Original"), OC.AUDIT_EP)
        n += 1
        if time.perf_counter() - t0 > budget_s:
            break
    dt = time.perf_counter() - t0
    return {"value": n * len(constraints) / dt, "unit": "evals/s", "cores": 1, "kind": "port",
            "sample": "%d of the same synthetic Pods x %d constraints, pure-Python oracle (tree-walking Rego "
                      "interpreter; the Go/OPA reference is not runnable here), %.1f s" % (n, len(constraints), dt)}


def cxx_host_rate(client, objs, nss, budget_s=5.0):
    """Orientation only (not the cpu_baseline): the engine's own host-side C++ evaluator -- the tree-walking Rego
    evaluation that renders messages for violating pairs -- run over EVERY (constraint, review) pair of a sample on one
    core, i.e. a compiled CPU execution of the same templates, without the match step."""
    from gatekeeper_amd import driver as D
    from gatekeeper_amd import synth
    sample = objs[:512]
    rins = [D.to_review_in(D.AugmentedUnstructured(D.Unstructured(o), synth.namespace_for(o, nss), "Original")) for o in sample]
    table = client.driver.engine.create_table(rins, keep_docs=True)
    cids = [client.driver.constraint_id(c) for c in client.constraints.values()]
    t0 = time.perf_counter()
    n = 0
    for r in range(len(sample)):
        for cid in cids:
            table.render(cid, r)
        n += len(cids)
        if time.perf_counter() - t0 > budget_s:
            break
    dt = time.perf_counter() - t0
    table.free()
    return {"value": n / dt, "unit": "evals/s", "cores": 1,
            "what": "gk_render (host C++ tree-walking evaluation of the template for one pair, incl. the ctypes call) over %d pairs" % n}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=50)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--reviews", type=int, default=100000, help="reviews per GPU")
    ap.add_argument("--config", type=int, default=1, choices=[1, 2], help="1: 30 PSP x Pods; 2: 50 constraints x mixed objects")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    args = ap.parse_args()

    import torch
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    dist = None
    if world > 1 or os.environ.get("GK_FORCE_DIST"):   # GK_FORCE_DIST=1: exercise the sharded path on one GPU (world size 1)
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        torch.cuda.set_device(local_rank)
        dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))
    assert torch.cuda.is_available(), "bench.py needs an MI355X (there is no CPU fallback)"
    dev = torch.device("cuda", local_rank)
    torch.cuda.set_device(dev)

    from gatekeeper_amd import driver as D
    from gatekeeper_amd import synth
    from gatekeeper_amd.sweep import ShardedSweep

    fx = synth.load_fixtures()
    templates = synth.psp_templates(fx)
    constraints = synth.psp_constraints() if args.config == 1 else synth.audit_constraints()
    nss = synth.gen_namespaces()
    # weak scaling: every rank owns `reviews` objects of the global, seeded object stream
    objs = synth.gen_objects(args.reviews, seed=synth.SEED + rank, mixed=(args.config == 2))

    drv = D.Driver(device=local_rank, hostemu=False)
    client = D.Client(drv)
    for t in templates:
        client.AddTemplate(t)
    for k in constraints:
        client.AddConstraint(k)
    sweep = ShardedSweep(client, objs, nss, dist=dist, device=dev)

    def barrier():
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize()

    if args.warmup:
        sweep.sweep(args.warmup)
    barrier()
    t0 = time.perf_counter()
    res = sweep.sweep(args.steps)          # exactly `steps` launches (+ exchanges when sharded) in the timed region
    barrier()
    dt = time.perf_counter() - t0
    counts = sweep.sweep(1, download=True).counts
    # isolated kernel duration: a second, untimed pass with one HIP event pair per launch (the timed region above
    # brackets all launches with one pair, i.e. its average includes the gaps between consecutive launches)
    os.environ["GK_EVENT_PER_LAUNCH"] = "1"
    iso = sweep.sweep(min(args.steps, 20))
    del os.environ["GK_EVENT_PER_LAUNCH"]
    if dist is not None:
        tmax = torch.tensor([dt], device=dev, dtype=torch.float64)
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        dt = float(tmax.item())

    if rank == 0:
        nc = len(constraints)
        evals = float(nc) * args.reviews * world * args.steps
        kernel_s = iso.fast_kernel_ms / 1e3          # average duration of the dominant kernel alone (per-launch events)
        achieved = res.algo_bytes / kernel_s / 1e9 if kernel_s > 0 else 0.0
        full_table_bytes = int(res.n_rows) * 16 + args.reviews * 4   # what a kernel streaming every row would read
        out = {
            "metric": "AdmissionReview x constraint evals/sec",
            "value": evals / dt, "unit": "evals/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": dt / args.steps * 1e3, "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": "u32", "data": "synthetic",
            "config": {"workload": ("configs[1]: 30 gatekeeper PSP constraints (5 in-tree PSP templates x 6 parameterisations) x "
                                    "%d synthetic Pod AdmissionReviews per GPU" % args.reviews) if args.config == 1 else
                       ("configs[2]: audit sweep, 50 constraints x %d mixed synthetic cluster objects per GPU" % args.reviews),
                       "constraints": nc, "reviews_per_gpu": args.reviews, "rows_per_gpu": int(res.n_rows), "rows_read_per_gpu": int(res.n_rows_read),
                       "parallelism": "objects sharded across %d GPU(s); one RCCL all-gather of [violation bitmaps | counts] per sweep" % world,
                       "violating_pairs_rank0": int(counts.sum())},
            "roofline": {"bound": "hbm", "kernel": "gk_eval_tiles", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": achieved / HBM_PEAK_GBS, "traffic": None, "algo_bytes_per_launch": int(res.algo_bytes),
                         "avg_kernel_ms": iso.fast_kernel_ms, "launches_timed": int(iso.n_launches),
                         "avg_launch_ms_back_to_back": res.fast_kernel_ms, "lds_bytes_per_tile": int(res.lds_bytes),
                         "full_table_bytes": full_table_bytes,
                         "full_table_GBps": full_table_bytes / kernel_s / 1e9 if kernel_s > 0 else None,
                         "kernel_only_evals_per_s": nc * args.reviews / kernel_s if kernel_s > 0 else None},
        }
        # HBM traffic of the dominant kernel from the committed rocprofv3 PMC passes of this same command (bench.py
        # cannot run a profiler around itself); only reported when the profiled workload is the one just timed
        try:
            pmc = json.load(open(os.path.join(ROOT, "profiles", "pmc_latest.json")))
            if pmc.get("config") == args.config and pmc.get("reviews") == args.reviews and world == 1:
                out["roofline"]["traffic"] = pmc["hbm_bytes_per_launch"]
                out["roofline"]["traffic_source"] = pmc["source"]
        except (OSError, ValueError):
            pass
        if not args.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline(templates, constraints, objs[:20000], nss)
            out["cpu_host_evaluator_cxx"] = cxx_host_rate(client, objs, nss)
        print(json.dumps(out))
    if dist is not None:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
