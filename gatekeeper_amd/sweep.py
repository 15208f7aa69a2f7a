"""Audit sweep over a resident, object-sharded set.

The reference's audit loop is serial (pkg/audit/manager.go:591-642: for obj { Client.Review }).  Here every rank (one
process per GPU) keeps its shard of the flattened object set resident in HBM and sweeps it with one kernel launch; the only
exchange step is issued by the ENGINE on the kernel's stream through its own RCCL communicator (gk_table_sweep_sharded):
ONE in-place ncclAllGather of every shard's [violation bitmap | counts | fail-closed counts] slot -- the int64 totals are
the sums over the gathered slot tails --, so that every rank ends with the full constraints x objects answer.  Objects never
move.  Back-to-back sweeps of a shard are enqueue-only: four enqueues per pass, no host round trip.

torch.distributed is plumbing only: it carries the RCCL unique id from rank 0 to the other ranks (and, in the CPU tests,
stands in for the collectives through the test-only emulation library), and gathers the few top-k candidate records of
the audit lists.
"""
from __future__ import annotations

import ctypes as C

import numpy as np

from . import _lib as L
from . import driver as D
from . import synth


class ShardedResult:
    """one sharded sweep: global totals + (optionally) the gathered bitmaps, as numpy views"""

    def __init__(self, lib, ptr):
        o = ptr.contents
        self.world, self.rank, self.nc, self.stride_tiles, self.slot_bytes = o.world, o.rank, o.n_constraints, o.stride_tiles, o.slot_bytes
        self.constraint_ids = np.ctypeslib.as_array(o.constraint_ids, (self.nc,)).copy() if self.nc else np.zeros(0, np.uint32)
        self.shard_reviews = np.ctypeslib.as_array(o.shard_reviews, (self.world,)).copy()
        self.totals = np.ctypeslib.as_array(o.totals, (self.nc,)).copy() if self.nc else np.zeros(0, np.int64)
        self.kernel_ms, self.fast_kernel_ms, self.n_overflow = o.kernel_ms, o.fast_kernel_ms, o.n_overflow
        # fail closed: autoreject pairs per constraint, reviews beyond the engine's limits, reviews HandleReview rejected -- over ALL shards
        self.err_totals = np.ctypeslib.as_array(o.err_totals, (self.nc,)).copy() if self.nc else np.zeros(0, np.int64)
        self.beyond_limits, self.not_evaluated = int(o.beyond_limits), int(o.not_evaluated)
        # the exchange step of this call: the all-gather's own duration (0: an enqueue-only pass was handed out), bytes received, overlap mode
        self.exchange_ms, self.exchange_overlapped, self.exchange_bytes_inbound = float(o.exchange_ms), bool(o.exchange_overlapped), int(o.exchange_bytes_inbound)
        self.d_gathered = o.d_gathered
        self.gathered = None
        if o.gathered:
            raw = np.ctypeslib.as_array(o.gathered, (self.world * self.slot_bytes // 8,)).copy().view(np.uint8).reshape(self.world, self.slot_bytes)
            self.gathered = raw
        lib.gk_shard_free(ptr)

    def bitmaps(self):
        """-> list over ranks of [nc][ceil(shard reviews / 64)] uint64 violation bitmaps (the shard's own words)"""
        out = []
        for r in range(self.world):
            bm = self.gathered[r, :self.nc * self.stride_tiles * 8].view(np.uint64).reshape(self.nc, self.stride_tiles)
            out.append(bm[:, :(int(self.shard_reviews[r]) + 63) // 64].copy())
        return out

    def counts(self):
        """[world][nc] violating objects per shard and constraint"""
        off = self.nc * self.stride_tiles * 8
        return np.stack([self.gathered[r, off:off + self.nc * 4].view(np.uint32) for r in range(self.world)])


class ShardedSweep:
    """One rank's shard of the audited objects, resident in HBM.  `dist`: an initialised torch.distributed (any backend);
    None = single process (plain launches, no exchange)."""

    def __init__(self, client, objs=None, namespaces=None, dist=None, device=None, table=None, n=None, keep_docs=False):
        """Either `objs` (+ their Namespace map) to flatten here, or an existing resident `table` of `n` reviews."""
        self.client = client
        self.dist = dist
        self.objs = objs
        if table is None:
            rins = [D.to_review_in(D.AugmentedUnstructured(D.Unstructured(o), synth.namespace_for(o, namespaces), "Original"))
                    for o in objs]
            table = client.driver.engine.create_table(rins, keep_docs=keep_docs, resident=True)   # the audit set stays on the GPU
            n = len(objs)
        self.table = table
        self.n = n
        self.nc = len(client.constraints)
        self._cb = None
        if dist is not None:
            self._join(dist)

    def _join(self, dist):
        """join the engine to the communicator of this job"""
        eng = self.client.driver.engine
        rank, world = dist.get_rank(), dist.get_world_size()
        if eng.hostemu:
            # TEST-ONLY: the CPU emulation has no RCCL; its collectives are callbacks into torch.distributed (gloo)
            import torch

            def gather(_ctx, buf, slot_bytes):
                raw = (C.c_uint8 * (world * slot_bytes)).from_address(buf)
                mine = torch.from_numpy(np.frombuffer(raw, np.uint8, slot_bytes, rank * slot_bytes).copy())
                parts = [torch.zeros(slot_bytes, dtype=torch.uint8) for _ in range(world)]
                dist.all_gather(parts, mine)
                np.frombuffer(raw, np.uint8)[:] = torch.cat(parts).numpy()

            def reduce(_ctx, buf, n):
                arr = np.ctypeslib.as_array(buf, (n,))
                t = torch.from_numpy(arr.copy())
                dist.all_reduce(t)
                arr[:] = t.numpy()

            self._cb = (L.HE_ALLGATHER(gather), L.HE_ALLREDUCE(reduce))
            eng._check(eng.lib.gk_comm_init_host(eng.handle, rank, world, self._cb[0], self._cb[1], None))
            return
        ident = [None]
        if rank == 0:
            buf = C.create_string_buffer(L.GK_COMM_ID_BYTES)
            eng._check(eng.lib.gk_comm_unique_id(buf))
            ident[0] = buf.raw
        dist.broadcast_object_list(ident, src=0)
        eng._check(eng.lib.gk_comm_init(eng.handle, ident[0], rank, world))

    def comm_info(self):
        """(rank, ranks) as the engine's communicator itself reports them (ncclCommUserRank / ncclCommCount)"""
        eng = self.client.driver.engine
        r, w = C.c_int32(-1), C.c_int32(-1)
        eng._check(eng.lib.gk_comm_info(eng.handle, C.byref(r), C.byref(w)))
        return int(r.value), int(w.value)

    def sweep(self, steps=1, download=False, strict=False, collect=False):
        """`steps` passes of the hot path over the resident shard.  Single process: the launches are enqueued back to back and
        collected once -> EvalResult.  Sharded: every pass is local evaluation + the engine's exchange step, enqueued back to
        back on the shard's stream (GK_SHARD_ENQUEUE); the last pass collects -> ShardedResult of the last pass.  A sharded result carries `beyond_limits` / `not_evaluated` / `err_totals` (summed over all shards):
        objects the totals and bitmaps say nothing about.  strict=True raises driver.LimitError / driver.ReviewFailure for
        them, as Client.AuditAggregate reports them for a single table (every rank raises: the counts are global).
        collect=True: ALL `steps` passes are enqueue-only and the answer of
        the last one is collected without a further sweep (GK_SHARD_COLLECT; the engine sweeps once more only when that pass left
        reviews to the large-capacity re-run)."""
        if self.dist is None:
            for _ in range(steps):
                self.table.launch()
            return self.table.eval(download=download, collect_only=True)
        eng = self.client.driver.engine
        res = None
        for k in range(steps + (1 if collect else 0)):
            if k < steps - 1 or (collect and k < steps):   # sweep + exchange enqueued back to back on the shard's stream; the last pass collects
                eng._check(eng.lib.gk_table_sweep_sharded(eng.handle, self.table.handle, L.GK_SHARD_ENQUEUE, None))
                continue
            out = C.POINTER(L.gk_shard_out)()
            flags = (L.GK_SHARD_DOWNLOAD if download else 0) | (L.GK_SHARD_COLLECT if collect else 0)
            eng._check(eng.lib.gk_table_sweep_sharded(eng.handle, self.table.handle, flags, C.byref(out)))
            res = ShardedResult(eng.lib, out)
        if strict and res is not None:
            if res.beyond_limits:
                raise D.LimitError("%d object(s) of the sharded set are beyond the engine's limits: review them on the CPU driver" % res.beyond_limits)
            if res.not_evaluated:
                raise D.ReviewFailure(-1, "%d object(s) of the sharded set were rejected by HandleReview" % res.not_evaluated)
        return res

    def audit_lists(self, limit=20, msg_size=256):
        """Cross-shard audit lists (pkg/audit/manager.go:112-203, 885-941): every rank selects and renders the `limit`
        smallest violations per constraint of ITS shard (device top-k, Client.AuditAggregate's machinery), the candidates
        are gathered and merged in SVQueue.Less order -- the global `limit` smallest are among the per-shard ones.
        Needs the shard's objects (constructed with objs=..., keep_docs=True).  -> {constraint key: [violation dict]}"""
        c = self.client
        ev = self.table.eval()
        top = self.table.topk(limit)
        row = {int(cid): i for i, cid in enumerate(ev.constraint_ids)}
        mine = {}
        for cid, (cons, ea, scoped) in c._active(D.AUDIT_EP).items():
            if cid not in row:
                continue
            reviews, overflow = top.get(cid, ([], False))
            if overflow:
                reviews = [int(r) for r in D.EvalResult.bits(ev.viol[row[cid]], ev.n_reviews)]
            cand = []
            for r in reviews:
                obj = self.objs[r]
                g, ver, k = D.obj_gvk(obj)
                for v in self.table.render(cid, r):
                    cand.append({"group": g, "version": ver, "kind": k, "namespace": (obj.get("metadata") or {}).get("namespace", "") or "",
                                 "name": (obj.get("metadata") or {}).get("name", "") or "", "message": D.truncate_string(v["msg"], msg_size),
                                 "enforcementAction": ea, "enforcementActions": scoped})
            cand.sort(key=_sv_key)
            mine[(cons.get("kind", ""), cons.get("apiVersion", ""), (cons.get("metadata") or {}).get("name", ""))] = cand[:limit]
        everyone = [mine]
        if self.dist is not None:
            everyone = [None] * self.dist.get_world_size()
            self.dist.all_gather_object(everyone, mine)
        merged = {}
        for part in everyone:
            for k, cand in part.items():
                merged.setdefault(k, []).extend(cand)
        return {k: sorted(v, key=_sv_key)[:limit] for k, v in merged.items()}


def _sv_key(x):
    """SVQueue.Less (pkg/audit/manager.go:118-138), byte-wise like Go string comparison"""
    return tuple(s.encode("utf-8") for s in (x["group"], x["version"], x["kind"], x["namespace"], x["name"], x["message"], x["enforcementAction"]))
