"""Plan specialisation in the background (kernels.hip jit_for): the first Query after a policy change is answered by the
bytecode kernel without waiting for hiprtc, the plan-specialised kernel takes over once its module is loaded, and identical
source text is served from the code-object cache.

The reference's webhook gives a review 3 s by default and turns a Driver.Query error into an HTTP 500
(/root/reference/pkg/webhook/policy.go:208-211); a 2 s compile inside the first Query after every constraint change does
not fit that budget."""
import ctypes as C
import os
import time

import pytest

from gatekeeper_amd import driver as D
from gatekeeper_amd import synth
from oracle import client as OC
from oracle import target as OT


def _cache_stats(lib):
    h, c = C.c_uint64(), C.c_uint64()
    lib.gk_jit_cache_stats(C.byref(h), C.byref(c))
    return h.value, c.value


def _expect(oc, o, nss):
    res = oc.review(OT.AugmentedUnstructured(OT.Unstructured(o), synth.namespace_for(o, nss), "Original"), OC.WEBHOOK_EP)
    return sorted((r.constraint["metadata"]["name"], r.msg) for r in res)


@pytest.mark.gpu
def test_first_query_after_a_policy_change_does_not_wait_for_the_compiler(fixtures, tmp_path):
    for k in ("GK_NO_JIT", "GK_JIT_STRICT", "GK_JIT_SYNC", "GK_HOSTEMU_JIT"):
        os.environ.pop(k, None)
    os.environ["GK_JIT_CACHE_DIR"] = str(tmp_path)
    try:
        drv = D.Driver(device=0, hostemu=False)
        c, oc = D.Client(drv), OC.Client()
        lib = drv.engine.lib
        templates, constraints = synth.psp_templates(fixtures), synth.psp_constraints()
        for t in templates:
            c.AddTemplate(t)
            oc.add_template(t)
        for k in constraints[:-1]:
            c.AddConstraint(k)
            oc.add_constraint(k)
        nss = synth.gen_namespaces()
        objs = synth.gen_objects(48, seed=77)
        rvs = [D.AugmentedUnstructured(D.Unstructured(o), synth.namespace_for(o, nss), "Original") for o in objs]

        def query(i):
            resp = drv.QueryMatching(D.TARGET_NAME, list(c.constraints.values()), rvs[i])
            return sorted((r.constraint["metadata"]["name"], r.msg) for r in resp.results)
        # warm the process (HIP context, allocator pool, the ahead-of-time kernels) on the first policy set
        assert query(0) == _expect(oc, objs[0], nss)
        lib.gk_jit_quiesce()
        _, compiles0 = _cache_stats(lib)
        # ---- the policy changes: the NEXT query must not pay for hiprtc
        c.AddConstraint(constraints[-1])
        oc.add_constraint(constraints[-1])
        t0 = time.perf_counter()
        first = query(1)
        dt = time.perf_counter() - t0
        assert first == _expect(oc, objs[1], nss)
        assert dt < 0.05, "first query after AddConstraint took %.1f ms" % (dt * 1e3)
        more = [query(i) for i in range(2, 10)]             # still (or already) correct while the build is running
        assert more == [_expect(oc, objs[i], nss) for i in range(2, 10)]
        lib.gk_jit_quiesce()                                  # the specialised module is loaded now
        _, compiles1 = _cache_stats(lib)
        # hiprtc ran in the background, never inside a query (a plan is re-specialised when new key paths give it new predicate
        # classes -- the first reviews of a fresh engine still grow the path dictionary -- so more than one build may have run)
        assert compiles1 >= compiles0 + 1, (compiles0, compiles1)
        after = [query(i) for i in range(10, 48)]
        assert after == [_expect(oc, objs[i], nss) for i in range(10, 48)]
        assert sum(len(x) for x in after) > 0
        # ---- the same policies again (constraint removed and re-added): identical source text -> code-object cache
        c.RemoveConstraint(constraints[-1])
        query(0)
        lib.gk_jit_quiesce()
        hits_a, compiles_a = _cache_stats(lib)
        c.AddConstraint(constraints[-1])
        assert query(1) == first
        lib.gk_jit_quiesce()
        hits_b, compiles_b = _cache_stats(lib)
        assert hits_b > hits_a and compiles_b - compiles_a <= 1, (hits_a, hits_b, compiles_a, compiles_b)   # served from the cache
        assert any(f.endswith(".co") for f in os.listdir(str(tmp_path)))     # ... and on disk for the next process
    finally:
        os.environ.pop("GK_JIT_CACHE_DIR", None)


@pytest.mark.gpu
def test_disk_cache_is_on_by_default_and_serves_a_restarted_engine(fixtures, tmp_path, monkeypatch):
    """No GK_JIT_CACHE_DIR: the code objects go to $XDG_CACHE_HOME/gkgpu-jit (here a directory of the test), named by source hash, gfx950
    and the hiprtc version.  After gk_jit_cache_drop_memory -- what a restarted process sees -- a second engine with the same policies
    and the same table loads every plan group's kernel from the files: no hiprtc run, the same answers.  A policy set of several plan
    groups starts all its builds before it waits for the first (they compile side by side)."""
    for k in ("GK_NO_JIT", "GK_JIT_CACHE_DIR", "GK_JIT_ASYNC"):
        monkeypatch.delenv(k, raising=False)
    monkeypatch.setenv("XDG_CACHE_HOME", str(tmp_path))
    monkeypatch.setenv("GK_JIT_STRICT", "1")
    templates, constraints = synth.corpus(fixtures, 140)        # three plan groups
    nss = synth.gen_namespaces()

    def sweep():
        drv = D.Driver(device=0, hostemu=False)
        c = D.Client(drv)
        for t in templates:
            c.AddTemplate(t)
        for k in constraints:
            c.AddConstraint(k)
        batch = synth.NativeBatch(drv.engine.lib, 9000, seed=synth.SEED, mixed=True, start=0, namespaces=nss)
        table = drv.engine.create_table_native(batch.reviews, 9000, keep_docs=False, resident=True, pruned=True)
        t0 = time.perf_counter()
        table.launch()
        ev = table.eval(download=True, collect_only=True)
        dt = time.perf_counter() - t0
        counts = [int(x) for x in ev.counts]
        groups = int(ev.n_plan_groups)
        table.free(); batch.free(); drv.engine.close()
        return drv.engine.lib, counts, groups, dt
    from gatekeeper_amd import _lib
    _lib.load(hostemu=False).gk_debug_set(b"group_max", 64)    # (several plan groups: the corpus is one plan since round 6)
    lib, counts_cold, groups, dt_cold = sweep()
    assert groups == 3
    cache = os.path.join(str(tmp_path), "gkgpu-jit")
    assert lib.gk_jit_cache_dir().decode() == cache
    files = [f for f in os.listdir(cache) if f.endswith(".co")]
    assert len(files) >= 3 and all(f.startswith("gk_gfx950_rtc") for f in files), files
    hits0, compiles0 = _cache_stats(lib)
    assert compiles0 >= 3
    lib.gk_jit_cache_drop_memory()
    _, counts_warm, _, dt_warm = sweep()
    hits1, compiles1 = _cache_stats(lib)
    assert counts_warm == counts_cold and sum(counts_cold) > 0
    assert compiles1 == compiles0 and hits1 >= hits0 + 3, (hits0, hits1, compiles0, compiles1)     # every group from its file
    assert dt_warm < dt_cold
    # switched off: no directory, no files
    monkeypatch.setenv("GK_JIT_CACHE_DIR", "off")
    lib.gk_debug_set(b"group_max", 0)
    assert lib.gk_jit_cache_dir() == b""
