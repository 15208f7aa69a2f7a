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


class _DeviceBytes:
    """A byte range in device memory owned by the engine, exposed through the CUDA array interface so that torch can
    alias it without a copy."""

    def __init__(self, ptr, nbytes):
        self.__cuda_array_interface__ = {"shape": (nbytes,), "typestr": "|u1", "data": (int(ptr), False), "version": 2}


class ShardedSweep:
    """One rank's shard of the audited objects, resident in HBM, plus the exchange buffers.

    Exchange step of a pass: ONE all-gather of `[violation bitmap | per-constraint counts]` (bytes) per rank, so every
    rank ends with all shards' bitmaps and counts; the global totals are the sum of the gathered counts (`total_counts`).
    On the device path nothing waits on the host between passes: the collective reads the engine's result buffer in
    place (bitmap and counts are one contiguous allocation, aliased as a torch tensor) and is ordered after the kernels
    by the stream; the host synchronises once, when the passes are collected."""

    def __init__(self, client, objs=None, namespaces=None, dist=None, device=None, table=None, n=None):
        """Either `objs` (+ their Namespace map) to flatten here, or an existing resident `table` of `n` reviews."""
        self.client = client
        self.dist = dist
        if table is None:
            rins = [D.to_review_in(D.AugmentedUnstructured(D.Unstructured(o), synth.namespace_for(o, namespaces), "Original"))
                    for o in objs]
            table = client.driver.engine.create_table(rins, keep_docs=False, resident=True)   # the audit set stays on the GPU
            n = len(objs)
        self.table = table
        self.n = n
        self.nc = len(client.constraints)
        self.n_tiles = (self.n + 63) // 64
        self.bm_bytes = self.nc * self.n_tiles * 8
        self.stage = self.gathered_raw = None
        self.on_device = device is not None and str(device).startswith("cuda")
        self._ptrs = None            # (d_viol, d_counts) of the table's result buffers, known after the first collect
        self._alias = None           # torch view of the engine's [bitmap | counts] buffer (device path)
        self._alias_ptr = None
        if dist is not None:
            import torch
            self.world = dist.get_world_size()
            self.stage = torch.zeros(self.bm_bytes + self.nc * 4, dtype=torch.uint8, device=device)
            self.gathered_raw = torch.zeros(self.world * self.stage.numel(), dtype=torch.uint8, device=device)

    # -- views of the gathered bytes ---------------------------------------------------------------------------------
    @property
    def gathered(self):
        """[world * n_constraints * n_tiles] int64: every rank's violation bitmap, rank-major."""
        import torch
        g = self.gathered_raw.view(self.world, -1)[:, :self.bm_bytes].contiguous()
        return g.view(torch.int64).reshape(-1)

    @property
    def total_counts(self):
        """[n_constraints] int32: violating objects per constraint over all shards."""
        import torch
        c = self.gathered_raw.view(self.world, -1)[:, self.bm_bytes:].contiguous().view(torch.int32)
        return c.reshape(self.world, self.nc).sum(0, dtype=torch.int32)

    def _exchange(self):
        self.dist.all_gather_into_tensor(self.gathered_raw, self.stage)

    def sweep(self, steps=1, download=False):
        """`steps` passes of the hot path over the resident shard.  Single GPU: the launches are enqueued back to back
        and collected once.  Sharded: every pass is followed by its exchange step.  Returns the EvalResult of the last
        pass (kernel time = average over the passes)."""
        if self.dist is None:
            for _ in range(steps):
                self.table.launch()
            return self.table.eval(download=download, collect_only=True)
        import torch
        ev = None
        if not self.on_device:
            # CPU path of the exchange (gloo; used by tests/test_sweep_dist.py with the test-only emulated kernels):
            # the same staging layout and collective on host tensors built from the downloaded results
            import numpy as np
            for _ in range(steps):
                ev = self.table.eval(download=True)
                raw = np.concatenate([ev.viol.reshape(-1).view(np.uint8), ev.counts.astype(np.int32).view(np.uint8)])
                self.stage.copy_(torch.from_numpy(raw.copy()))
                self._exchange()
            return ev
        done = 0
        if self._ptrs is None and steps > 0:
            # first pass ever: one collect to learn where the table's result buffers live (stable afterwards)
            self.table.launch()
            ev = self.table.eval(download=False, collect_only=True)
            self._ptrs = (ev.d_viol, ev.d_counts)
            self._stage_and_exchange()
            done = 1
        for _ in range(done, steps):
            self.table.launch()
            self._stage_and_exchange()
        if steps > done:
            ev = self.table.eval(download=False, collect_only=True)   # the one host synchronisation of the sweep
            self._ptrs = (ev.d_viol, ev.d_counts)
        if ev is not None and ev.n_overflow:
            # reviews that overflow the LDS element capacities are re-run by the big variant only when a pass is
            # collected, i.e. AFTER its exchange step: their bits would be missing from what the other ranks gathered
            raise RuntimeError("%d review(s) of this shard need the large-capacity kernel variant; the stream-ordered exchange "
                               "would gather their bits too early (create the table with resident=True so the plan variant fits "
                               "the shard's arrays)" % ev.n_overflow)
        torch.cuda.current_stream().synchronize()
        if download:
            self.table.launch()
            ev = self.table.eval(download=True, collect_only=True)
        return ev

    def _stage_and_exchange(self):
        d_viol, d_counts = self._ptrs
        if d_counts == d_viol + self.bm_bytes:
            # the engine keeps [bitmap | counts] contiguous: gather straight from its buffer, no staging copies
            if self._alias is None or self._alias_ptr != d_viol:
                import torch
                try:
                    self._alias = torch.as_tensor(_DeviceBytes(d_viol, self.stage.numel()), device=self.stage.device)
                except Exception:
                    self._alias = False
                self._alias_ptr = d_viol
            if self._alias is not False:
                self.dist.all_gather_into_tensor(self.gathered_raw, self._alias)
                return
        base = self.stage.data_ptr()
        _d2d(base, d_viol, self.bm_bytes)
        _d2d(base + self.bm_bytes, d_counts, self.nc * 4)
        self._exchange()
