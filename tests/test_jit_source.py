"""The text the product hands to hiprtc for a plan-specialised build of the dominant kernel (csrc/jit_source.hpp
assemble_jit_source: prelude + plan.hpp + vm_core.hpp + generated jit_row / jit_formulas + kernel_body.inc), produced by
the GPU-less test build for real plans, geometries and tables, and compiled HERE with the same hiprtc call the product
makes on the device (--offload-arch=gfx950 -O3 -std=c++17; hiprtc needs no GPU).  A plan whose generated source does not
compile for gfx950 -- a construct g++ accepts and the device compiler does not, a register / LDS budget the launch bounds
cannot meet -- fails in the build container instead of on the GPU box.  Scratch use of the compiled kernel is reported by
the code object's metadata and must be zero for the bench plan."""
import ctypes
import glob
import hashlib
import os

import pytest

from gatekeeper_amd import driver as D
from gatekeeper_amd import synth


def _hiprtc():
    for name in ("libhiprtc.so", "/opt/rocm/lib/libhiprtc.so"):
        try:
            return ctypes.CDLL(name)
        except OSError:
            continue
    return None


def compile_gfx950(rtc, src):
    """(ok, log, code) of hiprtcCompileProgram with the product's options (kernels.hip jit_build)"""
    prog = ctypes.c_void_p()
    assert rtc.hiprtcCreateProgram(ctypes.byref(prog), src.encode(), b"gk_plan.hip", 0, None, None) == 0
    opts = (ctypes.c_char_p * 3)(b"--offload-arch=gfx950", b"-O3", b"-std=c++17")
    rc = rtc.hiprtcCompileProgram(prog, 3, opts)
    n = ctypes.c_size_t()
    rtc.hiprtcGetProgramLogSize(prog, ctypes.byref(n))
    log = ctypes.create_string_buffer(max(n.value, 1))
    rtc.hiprtcGetProgramLog(prog, log)
    code = b""
    if rc == 0:
        rtc.hiprtcGetCodeSize(prog, ctypes.byref(n))
        buf = ctypes.create_string_buffer(n.value)
        rtc.hiprtcGetCode(prog, buf)
        code = buf.raw
    rtc.hiprtcDestroyProgram(ctypes.byref(prog))
    return rc == 0, log.value.decode(errors="replace"), code


def _dump_sources(monkeypatch, tmp_path, build_and_eval, env=()):
    monkeypatch.setenv("GK_HOSTEMU_KERNEL", "jit")
    monkeypatch.setenv("GK_EMU_HIP_SOURCE_DIR", str(tmp_path))
    monkeypatch.setenv("GK_EMU_GRID", "8")
    for k, v in env:
        monkeypatch.setenv(k, str(v))
    build_and_eval()
    texts = {}
    for f in sorted(glob.glob(os.path.join(str(tmp_path), "gk_plan_*.hip"))):
        t = open(f).read()
        texts.setdefault(hashlib.sha1(t.encode()).hexdigest(), (os.path.basename(f), t))
    assert texts, "the emulated evaluation produced no plan-specialised source"
    return list(texts.values())


def _bench_plan(n):
    def run():
        fx = synth.load_fixtures()
        drv = D.Driver(device=0, hostemu=True)
        client = D.Client(drv)
        for t in synth.psp_templates(fx):
            client.AddTemplate(t)
        for k in synth.audit_constraints():
            client.AddConstraint(k)
        batch = synth.NativeBatch(drv.engine.lib, n, seed=synth.SEED, mixed=True, start=0, namespaces=synth.gen_namespaces())
        table = drv.engine.create_table_native(batch.reviews, n, keep_docs=False, resident=True)
        table.launch()
        table.eval(download=True, collect_only=True)
    return run


def _scratch_bytes(code):
    """.private_segment_fixed_size of gk_jit_tiles from the code object's msgpack metadata (note record), -1 if not found"""
    i = code.find(b".private_segment_fixed_size")
    if i < 0:
        return -1
    v = code[i + len(b".private_segment_fixed_size")]
    if v < 0x80:
        return v                                   # msgpack positive fixint
    if v == 0xCC:
        return code[i + len(b".private_segment_fixed_size") + 1]
    if v == 0xCD:
        return int.from_bytes(code[i + len(b".private_segment_fixed_size") + 1:][:2], "big")
    if v == 0xCE:
        return int.from_bytes(code[i + len(b".private_segment_fixed_size") + 1:][:4], "big")
    return -1


def _row_load_wait_gaps(code, tmp_path):
    """per `global_load_dwordx4` of gk_jit_tiles: instructions on the fall-through path up to the next `s_waitcnt vmcnt`; None without llvm-objdump"""
    import re
    import subprocess
    objdump = "/opt/rocm/lib/llvm/bin/llvm-objdump"
    if not os.path.exists(objdump):
        return None
    co = os.path.join(str(tmp_path), "gk_plan.co")
    with open(co, "wb") as f:
        f.write(code)
    dis = subprocess.run([objdump, "-d", co], capture_output=True, text=True).stdout
    ops = [m.group(1) + " " + m.group(2) for m in (re.match(r"\s+([a-z_0-9]+)\s+(.*?)\s*//", line) for line in dis.splitlines()) if m]
    gaps = []
    for i, op in enumerate(ops):
        if op.startswith("global_load_dwordx4"):
            gap = 10 ** 6   # (an unconditional branch ends the fall-through path: what follows in the text is another block)
            for k in range(i + 1, len(ops)):
                if ops[k].startswith("s_waitcnt") and "vmcnt" in ops[k]:
                    gap = k - i
                    break
                if ops[k].startswith(("s_branch", "s_endpgm", "s_setpc")):
                    break
            gaps.append(gap)
    return gaps


@pytest.mark.parametrize("rpt", [64, 256])
def test_bench_plan_source_compiles_for_gfx950_without_scratch(monkeypatch, tmp_path, rpt):
    """configs[2]'s plan (50 constraints of the PSP family) in the geometries the bench tables use"""
    rtc = _hiprtc()
    if rtc is None:
        pytest.skip("libhiprtc.so is not installed")
    # (a 1 200-object table's plan variant has fewer accumulator words than the bench tables': more groups fit per CU and the budget
    #  would be 8 waves per SIMD = 64 VGPRs; the bench tables' budget is 6 waves = 80 VGPRs, which is what is checked here)
    env = [("GK_RPT", rpt)] + ([("GK_JIT_WAVES", 6)] if rpt == 256 else [])
    for name, text in _dump_sources(monkeypatch, tmp_path, _bench_plan(1200), env=env):
        ok, log, code = compile_gfx950(rtc, text)
        assert ok, "%s does not compile for gfx950:\n%s" % (name, log[-3000:])
        assert b"gk_jit_tiles" in code
        if rpt == 256:
            # the bench tables' geometry.  Round 3 packs the accumulators into 34 words per review: three 8-wave groups per CU =
            # 6 waves per SIMD = an 80-VGPR budget (it was 4 waves / 128 VGPRs and zero scratch).  At that budget the compiler
            # parks up to 16 dwords of per-thread invariants (the next item's review flags, two LDS addresses) in scratch: written
            # in the prologue / once per item, re-read in phase 2 -- never inside the chunk loop.  More than that is a regression.
            assert 0 <= _scratch_bytes(code) <= 96, "%s: the plan-specialised kernel spills (%d bytes of scratch per lane)" % (name, _scratch_bytes(code))
            # the prefetch distances the source intends are only real when the ISA shows them: no row load (16-byte loads) may be
            # waited for right behind its request.  (Round 5 found the next item's first rows waited for two instructions after the
            # request -- the compiler's copy into the chunk loop's entry registers -- with the whole output stage behind that wait.)
            gaps = _row_load_wait_gaps(code, tmp_path)
            if gaps is not None:
                assert gaps and min(gaps) >= 16, "%s: a row load is waited for %d instructions after its request (gaps %s)" % (name, min(gaps), gaps)


def test_bench_plan_text_sets_wave_priorities_and_carries_no_profiling_aids(monkeypatch, tmp_path):
    """round 6: the plan-specialised text defines a wave priority per phase (GK_PRIO_LEVELS 3003, wave 0's formula share one level up) and
    carries the phase marks / phase switches only when GK_KERNEL_PROF or GK_DBG_PHASE is set -- both measured on the device
    (profiles/r06_variants_{y,ad,ai}_*.log: -9 %, -4 %, -6 % on configs[2]); the ISA shows the s_setprio instructions and no clock reads"""
    import re
    import subprocess
    monkeypatch.delenv("GK_KERNEL_PROF", raising=False)
    monkeypatch.delenv("GK_DBG_PHASE", raising=False)
    monkeypatch.delenv("GK_JIT_PRIO", raising=False)
    texts = _dump_sources(monkeypatch, tmp_path, _bench_plan(1200), env=[("GK_RPT", 256)])
    rtc = _hiprtc()
    for name, text in texts:
        assert "#define GK_PRIO_LEVELS 3003" in text and "#define GK_PRIO_PART0 1" in text, name
        assert "#define GK_WITH_PROF" not in text, name
        if rtc is None:
            continue
        ok, log, code = compile_gfx950(rtc, text)
        assert ok, log[-3000:]
        objdump = "/opt/rocm/lib/llvm/bin/llvm-objdump"
        if os.path.exists(objdump):
            co = os.path.join(str(tmp_path), "prio.co")
            with open(co, "wb") as f:
                f.write(code)
            dis = subprocess.run([objdump, "-d", co], capture_output=True, text=True).stdout
            assert len(re.findall(r"\bs_setprio\b", dis)) >= 4, "no wave priorities in the ISA"
            # (the stagger of the launch word reads the clock in front of the group loop (three reads in the ISA); the seven phase marks would add seven reads)
            assert len(re.findall(r"s_memtime|s_memrealtime", dis)) <= 3, "clock reads (profiling marks) in the default text"
    # ... and with the marks asked for, they are in the text
    sub = tmp_path / "with_prof"
    sub.mkdir()
    monkeypatch.setenv("GK_KERNEL_PROF", "1")
    for name, text in _dump_sources(monkeypatch, sub, _bench_plan(1200), env=[("GK_RPT", 256)]):
        assert "#define GK_WITH_PROF" in text, name


def _pattern_plans():
    """the parity cases of the policy-compiler tests, run once more on the emulated plan-specialised kernel"""
    import test_library_patterns as L
    import test_pe_builtins as P
    import test_root_scope as R

    def run():
        L.test_library_patterns_one_plan("hostemu")
        L.test_library_patterns_second_batch("hostemu")
        L.test_library_patterns_third_batch("hostemu")
        R.test_values_compared_outside_iterations("hostemu")
        P.test_string_tests_on_iterated_keys("hostemu")
        P.test_definedness_of_opaque_builtin_results("hostemu")
    return run


def test_library_pattern_sources_compile_for_gfx950(monkeypatch, tmp_path):
    """every construct of the policy compiler that reaches generated code: dictionary predicates, element scopes with value
    slots, the root scope, key string tests, staged formulas of multi-group plans"""
    rtc = _hiprtc()
    if rtc is None:
        pytest.skip("libhiprtc.so is not installed")
    for name, text in _dump_sources(monkeypatch, tmp_path, _pattern_plans()):
        ok, log, _ = compile_gfx950(rtc, text)
        assert ok, "%s does not compile for gfx950:\n%s" % (name, log[-3000:])

@pytest.mark.parametrize("group_max", [0, 64], ids=["one-plan", "groups-of-64"])
def test_corpus_plan_sources_compile_for_gfx950_and_keep_their_prefetch_distance(monkeypatch, tmp_path, group_max):
    """the 200-template corpus -- as the ONE plan it is since round 6 (two banks of violation result slots) and as the plan groups of at most
    64 constraints of rounds 1-5: every text (128-review groups) through hiprtc; the next item's row loads are not waited for behind their
    request in any of them"""
    from gatekeeper_amd import _lib
    lib = _lib.load(hostemu=True)
    rtc = _hiprtc()
    if rtc is None:
        pytest.skip("libhiprtc.so is not installed")

    def run():
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
        table.eval(download=True, collect_only=True)
    assert lib.gk_debug_set(b"group_max", group_max) == 0
    try:
        texts = _dump_sources(monkeypatch, tmp_path, run)
    finally:
        lib.gk_debug_set(b"group_max", 0)
    assert len(texts) >= 3 if group_max else len(texts) == 1
    for name, text in texts:
        ok, log, code = compile_gfx950(rtc, text)
        assert ok, "%s does not compile for gfx950:\n%s" % (name, log[-3000:])
        # (the one-plan corpus -- 102 violation formulas, two banks of result registers -- at this test's 64-review geometry: 68 bytes; the
        #  geometry the device runs it at, 128-review groups, is measured there: profiles/INDEX_r06.md)
        assert 0 <= _scratch_bytes(code) <= 80, "%s: %d bytes of scratch per lane" % (name, _scratch_bytes(code))
        gaps = _row_load_wait_gaps(code, tmp_path)
        if gaps is not None:
            assert gaps and min(gaps) >= 16, "%s: a row load is waited for %d instructions after its request (gaps %s)" % (name, min(gaps), gaps)


def test_totals_rows_in_the_text_and_the_occupancy_guard(monkeypatch, tmp_path):
    """round 6, last step: the plan-specialised text of the benchmark plan keeps the per-constraint totals as one row per workgroup
    (`GK_TOT_K` = the constraints rounded up to 64: no popcount kernel behind a sweep) -- unless the LDS array would cost a resident
    row group per CU.  The cliff is measured (profiles/r06_variants_a{q,r,u}_*.log: 39 424 B per 256-review group keep four groups
    per CU, 39 432 B do not and the persistent grid takes 25 % longer): jit_source.hpp jit_tot_k decides with the exact footprint."""
    import subprocess
    monkeypatch.delenv("GK_FUSED_TOTALS", raising=False)
    for name, text in _dump_sources(monkeypatch, tmp_path, _bench_plan(1200), env=[("GK_RPT", 256)]):
        assert "#define GK_TOT_K 64\n" in text, name
        assert "GK_LDS_ADD(&s_tot[" in text and "out.partial[(size_t)blockIdx.x * GK_TOT_K + q] = s_tot[q]" in text, name
    csrc = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "gatekeeper_amd", "csrc")
    src = tmp_path / "tot_k.cpp"
    src.write_text('#include <cstdio>\n#include "plan.hpp"\n#include "codegen.hpp"\n#include "jit_source.hpp"\n'
                   'int main() { using namespace gk;\n'
                   '  // (constraints, accumulator bytes per group, reviews per group, threads, result words per half, list capacity)\n'
                   '  printf("%u %u %u %u %u\\n", jit_tot_k(50, 34816, 256, 512, 10, 256), jit_tot_k(100, 34816, 256, 512, 10, 256),\n'
                   '         jit_tot_k(100, 20000, 256, 512, 10, 256), jit_tot_k(300, 20000, 256, 512, 10, 256), jit_tot_k(200, 8192, 64, 256, 10, 128)); }\n')
    exe = tmp_path / "tot_k"
    subprocess.run(["g++", "-std=c++17", "-I", csrc, "-o", str(exe), str(src)], check=True)
    got = subprocess.run([str(exe)], capture_output=True, text=True, check=True).stdout.split()
    # configs[2]: 34 816 + 2 x 2 048 + 256 + 256 = 39 424 -> four groups, the array stays; 100 constraints would need 512 B -> three groups: popcount kernel;
    # smaller accumulators: room for it; beyond 256 constraints: never; 64-review groups are bound by waves, not LDS: stays
    assert got == ["64", "0", "128", "0", "256"], got
