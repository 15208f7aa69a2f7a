"""Multi-GPU path on CPU: world_size 2, gloo standing in for RCCL.  Each rank flattens and
evaluates its shard of the audit set through the engine's sharded-sweep entry point (gk_table_sweep_sharded: local
evaluation, then the ONE in-place all-gather of [bitmap | counts | tail] slots, the int64 totals being the sums over the gathered
tails -- the CPU emulation library takes the collective as a callback, the slot layout and the sequence are the product's).  Shards are UNEVEN
and not multiples of 64.  Every rank must end with the bitmaps, totals and merged top-k audit lists a single process gets."""
import os
import pickle
import socket

import numpy as np
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from gatekeeper_amd import driver as D
from gatekeeper_amd import synth
from gatekeeper_amd.sweep import ShardedSweep

SHARDS = [451, 333]   # uneven, not multiples of 64


def _client(policies="audit"):
    fx = synth.load_fixtures()
    c = D.Client(D.Driver(hostemu=True))
    # (the corpus tests are about SEVERAL plan groups -- one evaluation + exchange per group: groups of at most 64 constraints, as in
    #  rounds 1-5; since round 6 these corpora would be one plan.  Set in every process that builds a client, never reset: test processes)
    c.driver.engine.lib.gk_debug_set(b"group_max", 0 if policies == "audit" else 64)
    templates, constraints = (synth.psp_templates(fx), synth.audit_constraints()) if policies == "audit" else synth.corpus(fx, 82)
    for t, k in zip(templates, constraints) if policies != "audit" else ():
        if "ContainerLimits" in k["kind"]:   # (0.9 s of policy load per copy, three processes: the CPU suite's budget; 74 constraints are left)
            continue
        c.AddTemplate(t)
        c.AddConstraint(k)
    if policies == "audit":
        for t in templates:
            c.AddTemplate(t)
        for k in constraints:
            c.AddConstraint(k)
    return c


def _objs():
    objs = synth.gen_objects(sum(SHARDS), seed=21, mixed=True)
    objs[500] = dict(objs[3])   # the same object key on both shards: ties across shards in the merged lists
    return objs


def _worker(rank, world, port, out_dir, policies="audit"):
    dist.init_process_group("gloo", init_method="tcp://127.0.0.1:%d" % port, rank=rank, world_size=world)
    objs = _objs()
    lo = sum(SHARDS[:rank])
    shard = objs[lo:lo + SHARDS[rank]]
    sw = ShardedSweep(_client(policies), shard, synth.gen_namespaces(), dist=dist, device=torch.device("cpu"), keep_docs=True)
    sw.sweep(2)
    sw.sweep(2, collect=True)          # (GK_SHARD_COLLECT: the answer of the last enqueue-only pass; a set with several plan groups sweeps once more)
    res = sw.sweep(3, download=True)   # two enqueued sweep + exchange passes (GK_SHARD_ENQUEUE), the third collects
    lists = sw.audit_lists(limit=5)
    with open(os.path.join(out_dir, "rank_%d.pkl" % rank), "wb") as fh:
        pickle.dump({"bitmaps": res.bitmaps(), "totals": res.totals, "counts": res.counts(), "shards": res.shard_reviews, "lists": lists,
                     "ids": res.constraint_ids}, fh)
    dist.destroy_process_group()


import pytest   # noqa: E402


@pytest.mark.parametrize("policies", ["audit", "corpus72"])
def test_sharded_sweep_matches_single_process(tmp_path, policies):
    """corpus72: 74 templates / constraints = two plan groups (more than 64 distinct formulas): one evaluation + exchange
    per group, merged into one [constraints x objects] answer"""
    world = len(SHARDS)
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    mp.spawn(_worker, args=(world, port, str(tmp_path), policies), nprocs=world, join=True)
    objs = _objs()
    nss = synth.gen_namespaces()
    c = _client(policies)
    single = ShardedSweep(c, objs, nss, keep_docs=True)
    ref = single.table.eval()
    ref_lists = single.audit_lists(limit=5)
    n = len(objs)
    ref_bits = np.stack([np.unpackbits(ref.viol[r].view(np.uint8), bitorder="little")[:n] for r in range(ref.n_constraints)])
    for rank in range(world):
        got = pickle.load(open(os.path.join(str(tmp_path), "rank_%d.pkl" % rank), "rb"))
        assert list(got["shards"]) == SHARDS and (got["ids"] == ref.constraint_ids).all()
        bits = np.concatenate([np.stack([np.unpackbits(bm[r].view(np.uint8), bitorder="little")[:SHARDS[k]] for r in range(ref.n_constraints)])
                               for k, bm in enumerate(got["bitmaps"])], axis=1)
        assert (bits == ref_bits).all(), "rank %d sees a different global bitmap" % rank
        assert (got["totals"] == ref.counts.astype(np.int64)).all()                 # int64 totals over all shards (sums of the gathered slot tails)
        assert (got["counts"].sum(0) == ref.counts).all()                          # = sum of the gathered per-shard counts
        assert got["lists"] == ref_lists                                           # merged top-k == single-process LimitQueue
    assert ref.counts.sum() > 0 and sum(len(v) for v in ref_lists.values()) > 20


def _limit_objs():
    """two shards; each holds one object beyond the engine's limits (300 containers where predicates iterate elements) and one
    that makes a namespaceSelector constraint autoreject (namespaced object whose Namespace is not supplied)"""
    objs = synth.gen_objects(200, seed=5, mixed=True)
    for at in (7, 150):
        big = {"apiVersion": "v1", "kind": "Pod", "metadata": {"name": "huge-%d" % at, "namespace": "prod-1", "labels": {}},
               "spec": {"containers": [{"name": "c%d" % i, "image": "x", "securityContext": {"privileged": i == 299}} for i in range(300)]}}
        objs[at] = big
    return objs


def _limit_worker(rank, world, port, out_dir):
    dist.init_process_group("gloo", init_method="tcp://127.0.0.1:%d" % port, rank=rank, world_size=world)
    objs = _limit_objs()
    shard = objs[rank * 100:(rank + 1) * 100]
    sw = ShardedSweep(_client("audit"), shard, synth.gen_namespaces(), dist=dist, device=torch.device("cpu"), keep_docs=True)
    res = sw.sweep(1, download=True)
    raised = None
    try:
        sw.sweep(1, strict=True)
    except D.LimitError as ex:
        raised = str(ex)
    with open(os.path.join(out_dir, "lim_%d.pkl" % rank), "wb") as fh:
        pickle.dump({"beyond": res.beyond_limits, "not_evaluated": res.not_evaluated, "err_totals": res.err_totals, "totals": res.totals, "raised": raised}, fh)
    dist.destroy_process_group()


def test_sharded_sweep_fails_closed(tmp_path):
    """An object beyond the engine's limits contributes no bits on any rank: the sharded result must SAY so (round-2 advisor
    finding: the multi-GPU path was the last fail-open one), with the global count on every rank, and the autoreject pairs
    (match errors, one types.Result each in the reference) must be totalled like the violations."""
    world = 2
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    mp.spawn(_limit_worker, args=(world, port, str(tmp_path)), nprocs=world, join=True)
    objs = _limit_objs()
    single = ShardedSweep(_client("audit"), objs, synth.gen_namespaces(), keep_docs=True)
    # (the sharded exchange carries what the DEVICE answers: the reference here is the device's answer alone -- gk_table_eval's host
    #  evaluation of refused reviews, round 5, is switched off for it; with it on the same table has no refusal left)
    ref = single.table.eval(host_eval=False)
    assert sorted(int(r) for r in ref.too_big_reviews()) == [7, 150]
    full = single.table.eval()
    assert not full.too_big_reviews() and sorted(full.host_evaluated) == [7, 150]
    ref_err = np.array([int(np.unpackbits(ref.err[r].view(np.uint8)).sum()) for r in range(ref.n_constraints)], np.int64)
    for rank in range(world):
        got = pickle.load(open(os.path.join(str(tmp_path), "lim_%d.pkl" % rank), "rb"))
        assert got["beyond"] == 2 and got["not_evaluated"] == 0
        assert (got["err_totals"] == ref_err).all()
        assert (got["totals"] == ref.counts.astype(np.int64)).all()
        assert got["raised"] and "beyond the engine's limits" in got["raised"]


def _gpu_client():
    fx = synth.load_fixtures()
    c = D.Client(D.Driver(device=0, hostemu=False))
    for t in synth.psp_templates(fx):
        c.AddTemplate(t)
    for k in synth.audit_constraints():
        c.AddConstraint(k)
    return c


def _rccl_worker(rank, world, port, out_dir):
    os.environ["MASTER_ADDR"], os.environ["MASTER_PORT"] = "127.0.0.1", str(port)
    torch.cuda.set_device(0)
    dist.init_process_group("nccl", init_method="tcp://127.0.0.1:%d" % port, rank=rank, world_size=world, device_id=torch.device("cuda", 0))
    out = {}
    for name, objs in (("limits", _limit_objs()), ("plain", _plain_objs())):
        sw = ShardedSweep(_gpu_client(), objs, synth.gen_namespaces(), dist=dist, device=torch.device("cuda", 0), keep_docs=True)
        first = sw.sweep(1, download=True)
        # enqueue-only passes: the first runs directly, the second is captured into a graph, the others replay it.  collect=True hands
        # out the answer of the LAST REPLAY ("plain"); with reviews left to the large-capacity re-run ("limits": 300 containers) it
        # must fall back to one more, collecting sweep
        res = sw.sweep(6, download=True, collect=True)
        again = sw.sweep(3, download=True)
        last = sw.sweep(2, download=True, collect=True)
        out[name] = [{"bitmaps": r.bitmaps(), "totals": r.totals, "counts": r.counts(), "beyond": r.beyond_limits, "not_evaluated": r.not_evaluated,
                      "err_totals": r.err_totals, "kernel_ms": r.kernel_ms} for r in (first, res, again, last)]
    with open(os.path.join(out_dir, "rccl_%d.pkl" % rank), "wb") as fh:
        pickle.dump(out, fh)
    dist.destroy_process_group()


def _plain_objs():
    return synth.gen_objects(3000, seed=8, mixed=True)


@pytest.mark.gpu
@pytest.mark.parametrize("graph", ["direct", "graph", "overlap"])
def test_rccl_exchange_and_captured_sweeps_on_the_device(tmp_path, graph, monkeypatch):
    """The engine's own RCCL path on the MI355X at world size 1 (librccl through dlopen, the in-place ncclAllGather of the slots,
    the totals from the gathered tails), with the enqueue-only passes issued directly and replayed as a captured graph
    (GK_SHARD_GRAPH=1) or OVERLAPPED with the next pass's sweep (two slot buffers, exchange stream: what world size > 1 runs):
    bitmaps, totals and the fail-closed counts equal the plain evaluation of the same table, for the answers collected from the
    enqueue-only passes too."""
    monkeypatch.setenv("GK_SHARD_GRAPH", "1" if graph == "graph" else "0")   # (read by the engine in the spawned worker: opt-in captured replay)
    monkeypatch.setenv("GK_SHARD_OVERLAP", "1" if graph == "overlap" else "0")   # two slot buffers + exchange stream, the default at world size > 1
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    mp.spawn(_rccl_worker, args=(1, port, str(tmp_path)), nprocs=1, join=True)
    got_all = pickle.load(open(os.path.join(str(tmp_path), "rccl_0.pkl"), "rb"))
    for name, objs, beyond in (("limits", _limit_objs(), 2), ("plain", _plain_objs(), 0)):
        single = ShardedSweep(_gpu_client(), objs, synth.gen_namespaces(), keep_docs=True)
        ref = single.table.eval(host_eval=False)   # (the exchange carries the DEVICE's answer: the reference is the plain evaluation without the host evaluator's completions)
        n = len(objs)
        ref_err = np.array([int(np.unpackbits(ref.err[r].view(np.uint8)).sum()) for r in range(ref.n_constraints)], np.int64)
        ref_bits = np.stack([np.unpackbits(ref.viol[r].view(np.uint8), bitorder="little")[:n] for r in range(ref.n_constraints)])
        got = got_all[name]
        assert len(got) == 4 and ref.counts.sum() > 0 and len(ref.too_big_reviews()) == beyond
        for g in got:
            bits = np.stack([np.unpackbits(g["bitmaps"][0][r].view(np.uint8), bitorder="little")[:n] for r in range(ref.n_constraints)])
            assert (bits == ref_bits).all()
            assert (g["totals"] == ref.counts.astype(np.int64)).all() and (g["counts"][0] == ref.counts).all()
            assert g["beyond"] == beyond and g["not_evaluated"] == 0 and (g["err_totals"] == ref_err).all()
        # "plain": the collected answers came from the enqueue-only passes themselves (no collecting sweep ran: no kernel time reported)
        if name == "plain":
            assert got[1]["kernel_ms"] == 0 and got[3]["kernel_ms"] == 0 and got[2]["kernel_ms"] > 0


# ---- round 4: the OVERLAPPED exchange at world size 4.  The order of operations of the enqueue-only passes -- two slot buffers, the
# exchange stream, the "slot complete" / "exchange over" events -- is ONE definition (csrc/shard_pipe.hpp) with two backends: HIP
# streams + RCCL in the product, a worker thread + condition variables + the gloo callback in the CPU build.  Here it runs for real:
# pass k's all-gather travels on the exchange thread while pass k + 1 is evaluated, with three plan groups (three pipelines sharing
# one communicator: collectives in issue order) and uneven shards; every answer handed out equals the single-process one.
SHARDS4 = [173, 64, 201, 97]


def _client3():
    """> 128 distinct formulas = three plan groups; K8sContainerLimits left out (0.9 s of policy load per copy, four processes)"""
    fx = synth.load_fixtures()
    c = D.Client(D.Driver(hostemu=True))
    c.driver.engine.lib.gk_debug_set(b"group_max", 64)
    templates, constraints = synth.corpus(fx, 160)
    keep = [i for i, k in enumerate(constraints) if "ContainerLimits" not in k["kind"]]
    for i in keep:
        c.AddTemplate(templates[i])
        c.AddConstraint(constraints[i])
    return c


def _objs4():
    return synth.gen_objects(sum(SHARDS4), seed=33, mixed=True)


def _overlap_worker(rank, world, port, out_dir, policies):
    os.environ["GK_SHARD_OVERLAP"] = "1"
    dist.init_process_group("gloo", init_method="tcp://127.0.0.1:%d" % port, rank=rank, world_size=world)
    objs = _objs4()
    lo = sum(SHARDS4[:rank])
    shard = objs[lo:lo + SHARDS4[rank]]
    sw = ShardedSweep(_client3() if policies == "three-groups" else _client("audit"), shard, synth.gen_namespaces(), dist=dist, device=torch.device("cpu"))
    answers = [sw.sweep(1, download=True),                       # a collecting sweep
               sw.sweep(3, download=True),                       # two overlapped enqueue-only passes, the third collects
               sw.sweep(4, download=True, collect=True),         # four overlapped passes, the answer of the LAST one is handed out
               sw.sweep(2, download=True)]                       # ... and the buffers are sound afterwards
    with open(os.path.join(out_dir, "ovl_%d.pkl" % rank), "wb") as fh:
        pickle.dump([{"bitmaps": r.bitmaps(), "totals": r.totals, "counts": r.counts(), "shards": r.shard_reviews, "ids": r.constraint_ids,
                      "beyond": r.beyond_limits, "not_evaluated": r.not_evaluated} for r in answers], fh)
    dist.destroy_process_group()


@pytest.mark.parametrize("policies", ["one-group", "three-groups"])
def test_overlapped_exchange_at_world_size_four(tmp_path, policies):
    world = len(SHARDS4)
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    mp.spawn(_overlap_worker, args=(world, port, str(tmp_path), policies), nprocs=world, join=True)
    objs = _objs4()
    c = _client3() if policies == "three-groups" else _client("audit")
    single = ShardedSweep(c, objs, synth.gen_namespaces())
    ref = single.table.eval()
    n = len(objs)
    if policies == "three-groups":
        assert ref.n_plan_groups == 3
    ref_bits = np.stack([np.unpackbits(ref.viol[r].view(np.uint8), bitorder="little")[:n] for r in range(ref.n_constraints)])
    for rank in range(world):
        answers = pickle.load(open(os.path.join(str(tmp_path), "ovl_%d.pkl" % rank), "rb"))
        assert len(answers) == 4
        for k, got in enumerate(answers):
            assert list(got["shards"]) == SHARDS4 and (got["ids"] == ref.constraint_ids).all()
            bits = np.concatenate([np.stack([np.unpackbits(bm[r].view(np.uint8), bitorder="little")[:SHARDS4[j]] for r in range(ref.n_constraints)])
                                   for j, bm in enumerate(got["bitmaps"])], axis=1)
            assert (bits == ref_bits).all(), "rank %d, answer %d: a different global bitmap" % (rank, k)
            assert (got["totals"] == ref.counts.astype(np.int64)).all() and (got["counts"].sum(0) == ref.counts).all()
            assert got["beyond"] == 0 and got["not_evaluated"] == 0
    assert ref.counts.sum() > 100


# ---- world size 8 (the node SURVEY.md section 8e asks for), uneven shards -- one of them EMPTY, one a single object, one not a multiple
# of 64: the slot stride is the largest shard's, every rank ends with the same gathered bitmaps and totals, and the exchange figures
# the bench line prints (bytes received = 7 slots) are those of the slot layout
SHARDS8 = [130, 64, 1, 200, 0, 77, 128, 40]


def _world8_worker(rank, world, port, out_dir):
    dist.init_process_group("gloo", init_method="tcp://127.0.0.1:%d" % port, rank=rank, world_size=world)
    objs = synth.gen_objects(sum(SHARDS8), seed=37, mixed=True)
    lo = sum(SHARDS8[:rank])
    sw = ShardedSweep(_client("audit"), objs[lo:lo + SHARDS8[rank]], synth.gen_namespaces(), dist=dist, device=torch.device("cpu"))
    answers = [sw.sweep(1, download=True), sw.sweep(3, download=True, collect=True)]
    with open(os.path.join(out_dir, "w8_%d.pkl" % rank), "wb") as fh:
        pickle.dump([{"bitmaps": r.bitmaps(), "totals": r.totals, "counts": r.counts(), "shards": r.shard_reviews, "ids": r.constraint_ids, "slot": r.slot_bytes,
                      "stride": r.stride_tiles, "inbound": r.exchange_bytes_inbound, "exchange_ms": r.exchange_ms, "beyond": r.beyond_limits} for r in answers], fh)
    dist.destroy_process_group()


def test_uneven_shards_at_world_size_eight(tmp_path):
    world = len(SHARDS8)
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    os.environ["GK_HOST_THREADS"] = "1"
    try:
        mp.spawn(_world8_worker, args=(world, port, str(tmp_path)), nprocs=world, join=True)
    finally:
        del os.environ["GK_HOST_THREADS"]
    objs = synth.gen_objects(sum(SHARDS8), seed=37, mixed=True)
    single = ShardedSweep(_client("audit"), objs, synth.gen_namespaces())
    ref = single.table.eval()
    n = len(objs)
    ref_bits = np.stack([np.unpackbits(ref.viol[r].view(np.uint8), bitorder="little")[:n] for r in range(ref.n_constraints)])
    for rank in range(world):
        answers = pickle.load(open(os.path.join(str(tmp_path), "w8_%d.pkl" % rank), "rb"))
        for k, got in enumerate(answers):
            assert list(got["shards"]) == SHARDS8 and (got["ids"] == ref.constraint_ids).all()
            assert got["stride"] == (max(SHARDS8) + 63) // 64 and got["inbound"] == 7 * got["slot"]
            bits = np.concatenate([np.stack([np.unpackbits(bm[r].view(np.uint8), bitorder="little")[:SHARDS8[j]] for r in range(ref.n_constraints)]).reshape(ref.n_constraints, SHARDS8[j])
                                   for j, bm in enumerate(got["bitmaps"])], axis=1)
            assert (bits == ref_bits).all(), "rank %d, answer %d: a different global bitmap" % (rank, k)
            assert (got["totals"] == ref.counts.astype(np.int64)).all() and (got["counts"].sum(0) == ref.counts).all() and got["beyond"] == 0
        assert answers[0]["exchange_ms"] > 0 and answers[1]["exchange_ms"] == 0     # a collecting exchange is timed, a handed-out enqueue-only pass is not
    assert ref.counts.sum() > 100
