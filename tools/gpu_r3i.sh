#!/bin/bash
# Round-3 visit i: formulas as ONE stage (chained blocks, GK_JIT_CHAIN) and the output stage split between two waves per half
# (GK_OUT_ONE_WAVE restores the single-wave form), A/B on one box; per-phase clocks of the new default; quick parity subset
set -u
tag=${1:-r3i}
mkdir -p gpurun_out; export TMPDIR=/tmp
run() { env "$@" python bench.py --steps 100 --warmup 5 --lean 2>gpurun_out/${tag}.err | python -c "
import sys, json
d = json.loads(sys.stdin.read().strip().splitlines()[-1])
print('$*  avg_kernel_ms %.4f ms_per_step %.4f frac %.4f pairs %d' % (d['roofline']['avg_kernel_ms'], d['ms_per_step'], d['roofline']['frac'], d['config']['violating_pairs_rank0']))" | tee -a gpurun_out/${tag}_var.log; tail -1 gpurun_out/${tag}.err | grep -v amdgpu.ids; }
run GK_JIT_CHAIN=0 GK_JIT_DEFINES=GK_OUT_ONE_WAVE
run GK_JIT_CHAIN=1 GK_JIT_DEFINES=GK_OUT_ONE_WAVE
run GK_JIT_CHAIN=0
run GK_JIT_CHAIN=1
run GK_JIT_CHAIN=0 GK_JIT_DEFINES=GK_OUT_ONE_WAVE
run GK_JIT_CHAIN=1
GK_KERNEL_PROF=1 python bench.py --steps 20 --warmup 3 --lean 2>&1 >/dev/null | grep "gkgpu prof" | tail -2 | tee -a gpurun_out/${tag}_var.log
GK_JIT_STRICT=1 timeout 900 python -m pytest tests -m gpu -x -q -k "row_group or synthetic_parity or edge_cases or config1_demo or corpus or library or audit" 2>&1 | tail -3 | tee gpurun_out/${tag}_pytest_gpu.log
