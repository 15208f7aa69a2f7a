"""Audit sweep over a resident, object-sharded set.

The reference's audit loop is serial (pkg/audit/manager.go:591-642: for obj { Client.Review }).  Here every rank
(one process per GPU) keeps its shard of the flattened object set resident in HBM and sweeps it with one kernel
launch; the only exchange step is an RCCL all-gather of the per-shard violation bitmaps plus an all-reduce of the
per-constraint counts, so that every rank ends with the full constraints x objects answer.  Objects never move.
"""
from __future__ import annotations

import ctypes

from . import driver as D
from . import synth

_hip = None


def _d2d(dst_ptr, src_ptr, nbytes):
    """device-to-device copy between a raw HIP pointer handed out by the C ABI and a torch tensor (plumbing only)."""
    global _hip
    if _hip is None:
        _hip = ctypes.CDLL("libamdhip64.so")
        _hip.hipMemcpyAsync.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_size_t, ctypes.c_int, ctypes.c_void_p]
    rc = _hip.hipMemcpyAsync(dst_ptr, src_ptr, nbytes, 3, None)   # hipMemcpyDeviceToDevice on the default stream
    if rc != 0:
        raise RuntimeError("hipMemcpyAsync failed: %d" % rc)


class ShardedSweep:
    """One rank's shard of the audited objects, resident in HBM, plus the exchange buffers."""

    def __init__(self, client, objs, namespaces, dist=None, device=None):
        self.client = client
        self.dist = dist
        rins = [D.to_review_in(D.AugmentedUnstructured(D.Unstructured(o), synth.namespace_for(o, namespaces), "Original"))
                for o in objs]
        self.table = client.driver.engine.create_table(rins, keep_docs=False, resident=True)   # the audit set stays on the GPU
        self.n = len(objs)
        self.nc = len(client.constraints)
        self.n_tiles = (self.n + 63) // 64
        self.gathered = self.total_counts = None
        self.on_device = device is not None and str(device).startswith("cuda")
        if dist is not None:
            import torch
            w = dist.get_world_size()
            self.local_bm = torch.empty(self.nc * self.n_tiles, dtype=torch.int64, device=device)
            self.gathered = torch.empty(w * self.nc * self.n_tiles, dtype=torch.int64, device=device)
            self.total_counts = torch.empty(self.nc, dtype=torch.int32, device=device)

    def sweep(self, steps=1, download=False):
        """`steps` passes of the hot path over the resident shard.  Single GPU: the launches are enqueued back to back
        and collected once.  Sharded: every pass is followed by its exchange step (all-gather of bitmaps, all-reduce of
        counts).  Returns the EvalResult of the last pass (kernel time = average over the passes)."""
        if self.dist is None:
            for _ in range(steps):
                self.table.launch()
            return self.table.eval(download=download, collect_only=True)
        import torch
        ev = None
        if not self.on_device:
            # CPU path of the exchange (gloo; used by tests/test_sweep_dist.py with the test-only emulated kernels):
            # same collectives on host tensors built from the downloaded bitmaps
            import numpy as np
            for _ in range(steps):
                ev = self.table.eval(download=True)
                self.local_bm.copy_(torch.from_numpy(ev.viol.reshape(-1).view(np.int64).copy()))
                self.total_counts.copy_(torch.from_numpy(ev.counts.astype(np.int32)))
                self.dist.all_gather_into_tensor(self.gathered, self.local_bm)
                self.dist.all_reduce(self.total_counts)
            return ev
        for _ in range(steps):
            self.table.launch()
            ev = self.table.eval(download=False, collect_only=True)   # sync: bitmaps of THIS pass are complete
            _d2d(self.local_bm.data_ptr(), ev.d_viol, self.nc * self.n_tiles * 8)
            _d2d(self.total_counts.data_ptr(), ev.d_counts, self.nc * 4)
            torch.cuda.current_stream().synchronize()
            self.dist.all_gather_into_tensor(self.gathered, self.local_bm)
            self.dist.all_reduce(self.total_counts)
        if download:
            self.table.launch()
            ev = self.table.eval(download=True, collect_only=True)
        return ev
