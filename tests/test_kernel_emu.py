"""The dominant kernel's HIP source (gatekeeper_amd/csrc/kernel_body.inc) executed on the CPU by the TEST-ONLY kernel
emulator (tests/native/kernel_emu.hpp: one fiber per GPU thread, barriers and wave collectives with GPU semantics) and
compared, bit for bit, with the per-review evaluation of the same plan -- so the kernel's STRUCTURE (persistent
workgroups walking several row groups, host-built chunk lists, the double-buffered list staging, per-wave loop bounds,
result words in LDS, multi-pass groups, list overflow -> big variant) is checked in the GPU-less container, for the
generic bytecode build and for the plan-specialised build (the text hiprtc compiles on the device, here through g++).
`GK_HOSTEMU_KERNEL=1|jit` switches the check on for ANY hostemu evaluation (a mismatch raises); these cases pin the
geometries and limits the rest of the suite does not reach by itself."""
import pytest

from gatekeeper_amd import driver as D
from gatekeeper_amd import synth


def _sweep(monkeypatch, mode, n, env):
    monkeypatch.setenv("GK_HOSTEMU_KERNEL", mode)
    for k, v in env.items():
        monkeypatch.setenv(k, str(v))
    fx = synth.load_fixtures()
    drv = D.Driver(device=0, hostemu=True)
    client = D.Client(drv)
    for t in synth.psp_templates(fx):
        client.AddTemplate(t)
    for k in synth.audit_constraints():
        client.AddConstraint(k)
    nss = synth.gen_namespaces()
    batch = synth.NativeBatch(drv.engine.lib, n, seed=synth.SEED, mixed=True, start=0, namespaces=nss)
    table = drv.engine.create_table_native(batch.reviews, n, keep_docs=False, resident=True)
    table.launch()
    ev = table.eval(download=True, collect_only=True)     # raises EngineError when the emulated kernel and the per-review path differ
    assert int(ev.counts.sum()) > 0
    return ev


@pytest.mark.parametrize("mode", ["1", "jit"])
@pytest.mark.parametrize("env", [
    {"GK_RPT": 64, "GK_EMU_GRID": 8},                              # 4-wave groups, several groups per persistent workgroup
    {"GK_RPT": 256, "GK_EMU_GRID": 8},                             # 8-wave groups, result words alias the list buffer (jit)
    {"GK_RPT": 256, "GK_FORCE_RPP": 64, "GK_EMU_GRID": 8},         # multi-pass groups: four passes over the same chunk list
    {"GK_RPT": 512, "GK_FORCE_RPP": 128, "GK_EMU_GRID": 8},        # 16 waves, two passes
    {"GK_RPT": 128, "GK_EMU_LIST_CAP": 24, "GK_EMU_GRID": 8},      # lists overflow: the groups' reviews take the big variant
    {"GK_RPT": 64, "GK_EMU_GRID": 32, "GK_EMU_STAGGER": 1},        # staggered grid, 4 workgroups per XCD over 24 groups: 3 per XCD = a partial FIRST round only
    {"GK_RPT": 64, "GK_EMU_GRID": 16, "GK_EMU_STAGGER": 1},        # 2 per XCD: one full round + a partial one
    {"GK_RPT": 64, "GK_EMU_GRID": 32, "GK_EMU_STAGGER": 1, "N": 2900},   # 46 groups, 6 per XCD over 4 workgroups: one full round + a partial one of 2
], ids=["rpt64", "rpt256", "rpt256-4pass", "rpt512-2pass", "list-overflow", "stagger-partial-first", "stagger-partial-last", "stagger-partial-last-4"])
def test_kernel_source_on_the_emulator(monkeypatch, mode, env):
    _sweep(monkeypatch, mode, env.get("N", 1500), {k: v for k, v in env.items() if k != "N"})


@pytest.mark.parametrize("group_max", [0, 64], ids=["one-plan", "groups-of-64"])
def test_corpus_plan_groups_on_the_emulator(monkeypatch, group_max):
    """The 200-template corpus.  one-plan (round 6): ONE plan of 102 distinct violation formulas -- two banks of 64 violation result slots
    (jit_source.hpp jit_res_macros: kinds 0 and 3, a register pair each; the output stage gathers a constraint's word from the bank its
    slot lives in), constraints beyond the first 64 read their slots from the plan's slot table.  groups-of-64: the four plan groups of
    rounds 1-5 (the path a set beyond 256 formulas still takes), up to 64 result slots per kind -- lanes 32..63 of the result registers,
    element scopes read at use with rolling registers.  128-review groups; the plan-specialised text on the emulator against the
    per-review evaluation."""
    from gatekeeper_amd import _lib
    lib = _lib.load(hostemu=True)
    assert lib.gk_debug_set(b"group_max", group_max) == 0
    monkeypatch.setenv("GK_HOSTEMU_KERNEL", "jit")
    monkeypatch.setenv("GK_EMU_GRID", "8")
    fx = synth.load_fixtures()
    templates, constraints = synth.corpus(fx, 200)
    drv = D.Driver(device=0, hostemu=True)
    client = D.Client(drv)
    for t in templates:
        client.AddTemplate(t)
    for k in constraints:
        client.AddConstraint(k)
    n = 512
    batch = synth.NativeBatch(drv.engine.lib, n, seed=synth.SEED, mixed=True, start=0, namespaces=synth.gen_namespaces())
    table = drv.engine.create_table_native(batch.reviews, n, keep_docs=False, resident=True)
    table.launch()
    try:
        ev = table.eval(download=True, collect_only=True)     # raises EngineError when the emulated kernel and the per-review path differ
    finally:
        lib.gk_debug_set(b"group_max", 0)
    assert (int(ev.n_plan_groups) >= 3 if group_max else int(ev.n_plan_groups) == 1) and int(ev.counts.sum()) > 1000 and ev.n_constraints == 200
