"""Audit sweep over a resident, object-sharded set.

The reference's audit loop is serial (pkg/audit/manager.go:591-642: for obj { Client.Review }).  Here every rank
(one process per GPU) keeps its shard of the flattened object set resident in HBM and sweeps it with one kernel
launch; the only exchange step is an RCCL all-gather of the per-shard violation bitmaps plus an all-reduce of the
per-constraint counts, so that every rank ends with the full constraints x objects answer.  Objects never move.
"""
from __future__ import annotations

import ctypes

import numpy as np

from . import driver as D
from . import synth


class ShardedSweep:
    def __init__(self, client, objs, namespaces, dist=None, device=None):
        self.client = client
        self.dist = dist
        self.device = device
        rins = [D.to_review_in(D.AugmentedUnstructured(D.Unstructured(o), synth.namespace_for(o, namespaces), "Original")) for o in objs]
        self.table = client.driver.engine.create_table(rins, keep_docs=False)
        self.n = len(objs)
        self.nc = len(client.constraints)
        self.n_tiles = (self.n + 63) // 64
        self._pending = 0
        self.gathered = None
        self.total_counts = None
        if dist is not None:
            import torch
            w = dist.get_world_size()
            self.local_bm = torch.empty(self.nc * self.n_tiles, dtype=torch.int64, device=device)
            self.gathered = torch.empty(w * self.nc * self.n_tiles, dtype=torch.int64, device=device)
            self.total_counts = torch.empty(self.nc, dtype=torch.int32, device=device)

    def step(self):
        """one pass of the hot path over the resident shard (+ the exchange step when sharded across GPUs)"""
        if self.dist is None:
            self.table.launch()
            self._pending += 1
            return
        import torch
        # the exchange needs the device-resident results of THIS launch: collect, then hand the raw device pointers
        # to RCCL through zero-copy torch views
        ev = self.table.eval(download=False)
        self._last = ev
        nbytes = self.nc * self.n_tiles * 8
        hip = torch.cuda.current_stream().cuda_stream  # noqa: F841  (default stream: same one the engine launches on)
        src = (ctypes.c_char * nbytes).from_address(0)  # placeholder type for clarity
        del src
        torch.cuda.synchronize()
        _copy_from_device_ptr(self.local_bm, ev.d_viol, nbytes)
        _copy_from_device_ptr(self.total_counts, ev.d_counts, self.nc * 4)
        self.dist.all_gather_into_tensor(self.gathered, self.local_bm)
        self.dist.all_reduce(self.total_counts)

    def collect(self):
        if self.dist is None:
            ev = self.table.eval(download=True) if self._pending == 0 else self._collect_pending()
            self._pending = 0
            return ev
        return self.table.eval(download=True)

    def _collect_pending(self):
        # eval() = one more launch + finish: callers that want exactly K timed launches issue K-1 step() calls
        return self.table.eval(download=True)


def _copy_from_device_ptr(dst_tensor, src_ptr, nbytes):
    """device-to-device copy from a raw HIP pointer handed out by the C ABI into a torch tensor (plumbing only)."""
    import torch
    lib = ctypes.CDLL("libamdhip64.so")
    lib.hipMemcpyAsync.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_size_t, ctypes.c_int, ctypes.c_void_p]
    rc = lib.hipMemcpyAsync(dst_tensor.data_ptr(), src_ptr, nbytes, 3, None)
    assert rc == 0, "hipMemcpyAsync failed: %d" % rc
    torch.cuda.synchronize()
