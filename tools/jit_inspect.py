#!/usr/bin/env python3
"""Offline view of a plan-specialised kernel: puts the EXACT product text (dumped with GK_HOSTEMU_KERNEL=jit
GK_EMU_HIP_SOURCE_DIR=<dir>, or GK_JIT_DUMP=<file> on a GPU box) through hiprtc with the product's options -- hiprtc needs
no GPU -- and prints what bounds occupancy and issue: VGPR / SGPR / spills / LDS / scratch, instruction mix of the whole
kernel, and optionally the disassembly.
usage: tools/jit_inspect.py <gk_plan_*.hip> [--asm out.s] [--define NAME=VAL ...] [--sub OLD NEW] [--body kernel_body.inc] [--waits]
--body replaces the kernel body inside the dumped text by another file (what GK_JIT_BODY_FILE does on a GPU box);
--waits prints the kernel's vector-memory instructions, barriers and s_waitcnt vmcnt in program order: where a load is issued and
where the wave first waits for it -- the prefetch distances the source intends are only real when this listing shows them."""
import argparse
import collections
import os
import re
import subprocess
import sys
import time

ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
from test_jit_source import _hiprtc, compile_gfx950  # noqa: E402

OBJDUMP = "/opt/rocm/lib/llvm/bin/llvm-objdump"
READELF = "/opt/rocm/lib/llvm/bin/llvm-readelf"


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("src")
    ap.add_argument("--asm")
    ap.add_argument("--define", action="append", default=[])
    ap.add_argument("--sub", nargs=2, action="append", default=[])
    ap.add_argument("--body")
    ap.add_argument("--waits", action="store_true")
    a = ap.parse_args()
    text = open(a.src).read()
    if a.body:
        mark = "// Kernel bodies shared by the ahead-of-time build"
        at = text.index(mark)
        body = "".join(l for l in open(a.body) if not l.startswith("#include") and not l.startswith("#pragma once"))
        text = text[:at] + body + "}\n"
    for d in a.define:
        k, _, v = d.partition("=")
        text, n = re.subn(r"#define %s\b.*" % re.escape(k), "#define %s %s" % (k, v), text, count=1)
        assert n, "no #define %s in the text" % k
    for old, new in a.sub:
        assert old in text, old
        text = text.replace(old, new)
    rtc = _hiprtc()
    t0 = time.time()
    ok, log, code = compile_gfx950(rtc, text)
    dt = time.time() - t0
    if not ok:
        print(log[-4000:])
        sys.exit(1)
    co = (a.asm or "/tmp/jit_inspect") + ".co"
    open(co, "wb").write(code)
    notes = subprocess.run([READELF, "--notes", co], capture_output=True, text=True).stdout
    keys = (".name:", ".vgpr_count", ".agpr_count", ".sgpr_count", "spill_count", ".group_segment_fixed_size", ".private_segment_fixed_size", ".max_flat_workgroup_size")
    cur = {}
    for line in notes.splitlines():
        s = line.strip()
        if any(k in s for k in keys):
            k, _, v = s.lstrip("- ").partition(":")
            cur[k.strip()] = v.strip()
    print("compile %.1f s, code object %d B: %s" % (dt, len(code), " ".join("%s=%s" % kv for kv in sorted(cur.items()))))
    dis = subprocess.run([OBJDUMP, "-d", co], capture_output=True, text=True).stdout
    mix = collections.Counter()
    names = collections.Counter()
    for line in dis.splitlines():
        m = re.match(r"\s+([a-z_0-9]+)\s", line)
        if not m:
            continue
        op = m.group(1)
        names[op] += 1
        if op.startswith("s_cbranch") or op == "s_branch":
            mix["branch"] += 1
        elif op.startswith("s_waitcnt") or op == "s_nop":
            mix["wait/nop"] += 1
        elif op == "s_barrier":
            mix["barrier"] += 1
        elif op.startswith("s_"):
            mix["salu"] += 1
        elif op.startswith("v_readlane") or op.startswith("v_writelane") or op.startswith("v_readfirstlane"):
            mix["lane"] += 1
        elif op.startswith("v_accvgpr"):
            mix["accvgpr"] += 1
        elif op.startswith("v_"):
            mix["valu"] += 1
        elif op.startswith("ds_"):
            mix["lds"] += 1
        elif op.startswith("global_") or op.startswith("buffer_") or op.startswith("flat_") or op.startswith("scratch_"):
            mix["vmem"] += 1
        else:
            mix["other"] += 1
    print("static mix: " + " ".join("%s=%d" % kv for kv in mix.most_common()) + " total=%d" % sum(mix.values()))
    if a.waits:
        n = 0
        for line in dis.splitlines():
            m = re.match(r"\s+([a-z_0-9]+)\s+(.*?)\s*//", line)
            if not m:
                continue
            n += 1
            op = m.group(1)
            if op.startswith(("global_", "scratch_", "buffer_", "flat_")) or op == "s_barrier" or (op == "s_waitcnt" and "vmcnt" in m.group(2)):
                print("%6d  %s %s" % (n, op, m.group(2)))
    if a.asm:
        open(a.asm, "w").write(dis)
        print("disassembly -> %s" % a.asm)


if __name__ == "__main__":
    main()
