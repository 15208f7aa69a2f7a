"""Multi-GPU path on CPU: world_size 2, gloo.  Objects are block-sharded across ranks, each rank sweeps its shard and
the per-shard violation bitmaps are all-gathered together with the per-constraint counts (gatekeeper_amd/sweep.py).
The gathered result must equal the single-process sweep over all objects, bit for bit."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from gatekeeper_amd import driver as D
from gatekeeper_amd import synth
from gatekeeper_amd.sweep import ShardedSweep

N_PER_RANK = 256   # multiple of 64 so shard bitmaps concatenate word-aligned


def _client():
    fx = synth.load_fixtures()
    c = D.Client(D.Driver(hostemu=True))
    for t in synth.psp_templates(fx):
        c.AddTemplate(t)
    for k in synth.audit_constraints():
        c.AddConstraint(k)
    return c


def _worker(rank, world, port, out_dir):

    dist.init_process_group("gloo", init_method="tcp://127.0.0.1:%d" % port, rank=rank, world_size=world)
    objs = synth.gen_objects(N_PER_RANK * world, seed=21, mixed=True)
    shard = objs[rank * N_PER_RANK:(rank + 1) * N_PER_RANK]
    sw = ShardedSweep(_client(), shard, synth.gen_namespaces(), dist=dist, device=torch.device("cpu"))
    sw.sweep(2)
    np.save(os.path.join(out_dir, "gathered_%d.npy" % rank), sw.gathered.numpy())
    np.save(os.path.join(out_dir, "counts_%d.npy" % rank), sw.total_counts.numpy())
    dist.destroy_process_group()


def test_sharded_sweep_matches_single_process(tmp_path):
    world = 2
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    mp.spawn(_worker, args=(world, port, str(tmp_path)), nprocs=world, join=True)

    objs = synth.gen_objects(N_PER_RANK * world, seed=21, mixed=True)
    nss = synth.gen_namespaces()
    c = _client()
    rins = [D.to_review_in(D.AugmentedUnstructured(D.Unstructured(o), synth.namespace_for(o, nss), "Original")) for o in objs]
    ref = c.driver.engine.create_table(rins, keep_docs=False).eval()
    nc, nt = ref.n_constraints, N_PER_RANK // 64
    for rank in range(world):
        g = np.load(os.path.join(str(tmp_path), "gathered_%d.npy" % rank)).view(np.uint64).reshape(world, nc, nt)
        full = np.concatenate([g[r] for r in range(world)], axis=1)
        assert (full == ref.viol).all(), "rank %d sees a different global bitmap" % rank
        counts = np.load(os.path.join(str(tmp_path), "counts_%d.npy" % rank))
        assert (counts == ref.counts.astype(np.int32)).all()
    assert ref.counts.sum() > 0
