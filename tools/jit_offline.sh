#!/bin/bash
# Offline view of the plan-specialised kernel: assembles the same source text kernels.hip hands to hiprtc (generated
# part from GK_PLAN_SOURCE_DUMP) and compiles it with hipcc for gfx950, printing register / LDS / occupancy figures.
# (The EXACT text of a product build -- prelude, launch bounds and tuning defines as csrc/jit_source.hpp assembles them -- is
# written by the GPU-less test build with GK_HOSTEMU_KERNEL=jit GK_EMU_HIP_SOURCE_DIR=<dir>, or on a GPU box with
# GK_JIT_DUMP=<file>; tests/test_jit_source.py puts that text through hiprtc.  This script is the quick variant for A/B-ing
# geometry / prefetch / body variants by hand.)
# usage: [RPT=512 RPP=512 WAVES=4] tools/jit_offline.sh /tmp/plan_src.hip [outdir]   (dump the source with GK_PLAN_SOURCE_DUMP=... GK_PLAN_SOURCE_PARTS=2 for RPT >= 128)
set -e
gen=${1:-/tmp/plan_src.hip}; out=${2:-/tmp/jit_offline}; mkdir -p $out
here=$(cd $(dirname $0)/.. && pwd)
python3 - "$gen" "$out/gk_plan.hip" "$here/gatekeeper_amd/csrc" <<'PY'
import sys
gen, dst, src = sys.argv[1:4]
def text(name):
    return "".join(l for l in open(src + "/" + name) if not l.startswith("#include") and not l.startswith("#pragma once"))
import os
waves = os.environ.get("WAVES")
rpt = int(os.environ.get("RPT", "64")); rpp = int(os.environ.get("RPP", str(rpt))); block = int(os.environ.get("BLOCK", "0")) or (256 if rpt <= 128 else rpt * 2)
s = ("#include <hip/hip_runtime.h>\n" + ("#define GK_TILES_BOUNDS __launch_bounds__(%d, %s)\n" % (block, waves) if waves else "") + "#define GK_RPT_K %d\n#define GK_RPP_K %d\n#define GK_SKIP_BIG\n" % (rpt, rpp) + ("#define GK_BLOCK_K %s\n" % os.environ["BLOCK"] if os.environ.get("BLOCK") else "") + "".join("#define %s\n" % x.replace("=", " ") for x in os.environ.get("DEFINES", "").split(";") if x) + ("#define GK_PREFETCH %s\n" % os.environ["PREFETCH"] if os.environ.get("PREFETCH") else "") + text("plan.hpp") + text("vm_core.hpp") +
     "#define GK_RES_PROLOGUE const bool gk_l0 = __builtin_amdgcn_mbcnt_hi(~0u, __builtin_amdgcn_mbcnt_lo(~0u, 0u)) == 0u;\n"
     "#define GK_RES(kind, slot, b) do { const unsigned long long m_ = __ballot((b) != 0u); if (gk_l0) masks[(kind) * GK_RES_K + (slot)] = m_; } while (0)\n" + open(gen).read() +
     "namespace gk {\n#define GK_KERNEL_TILES gk_jit_tiles\n#define GK_KERNEL_BIG gk_jit_big\n#define GK_KERNEL_LINKAGE extern \"C\"\n"
     "#define GK_ROW_FN(r, i, ent, h, pv, heap, acc, on) jit_row(r, ent, h, heap, acc, on)\n#define GK_BIND_ALWAYS_STR 0\n"
     "#define GK_FORMULA_FN(pv, acc, flags, rows, heap, bounds) jit_formulas(pv, acc, flags, rows, heap, bounds)\n" +
     (text("kernel_body.inc") if not os.environ.get("BODY") else "".join(l for l in open(os.environ["BODY"]) if not l.startswith("#include") and not l.startswith("#pragma once"))) + "}\n")
open(dst, "w").write(s)
PY
cd $out
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -c gk_plan.hip -o gk_plan.o -Rpass-analysis=kernel-resource-usage --save-temps 2>&1 | grep -E "Function Name|VGPRs:|SGPRs:|ScratchSize|Occupancy|LDS Size|error" | sed 's/.*remark: //'
ls -la $out/*.s 2>/dev/null | head -3
