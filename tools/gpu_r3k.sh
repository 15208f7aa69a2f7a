#!/bin/bash
# Round-3 visit k: cost model of the formula shares (GK_JIT_LOOP_WEIGHT: a loop body's weight relative to straight-line code decides
# how the blocks are dealt to the two waves of a half), the corpus with / without chained formulas, referential tests on the device
set -u
tag=${1:-r3k}
mkdir -p gpurun_out; export TMPDIR=/tmp
run() { env "$@" python bench.py --steps 100 --warmup 5 --lean 2>gpurun_out/${tag}.err | python -c "
import sys, json
d = json.loads(sys.stdin.read().strip().splitlines()[-1])
print('$*  avg_kernel_ms %.4f ms_per_step %.4f frac %.4f pairs %d' % (d['roofline']['avg_kernel_ms'], d['ms_per_step'], d['roofline']['frac'], d['config']['violating_pairs_rank0']))" | tee -a gpurun_out/${tag}_var.log; tail -1 gpurun_out/${tag}.err | grep -v amdgpu.ids; }
run GK_JIT_LOOP_WEIGHT=3
run GK_JIT_LOOP_WEIGHT=1
run GK_JIT_LOOP_WEIGHT=2
run GK_JIT_LOOP_WEIGHT=3
run GK_JIT_LOOP_WEIGHT=1
run GK_JIT_LOOP_WEIGHT=2
runc() { env "$@" python bench.py --config 4 --steps 50 --warmup 5 --lean 2>gpurun_out/${tag}.err | python -c "
import sys, json
d = json.loads(sys.stdin.read().strip().splitlines()[-1])
print('config4 $*  ms_per_step %.4f kernel_ms_sum %.4f' % (d['ms_per_step'], d['roofline']['avg_kernel_ms']))" | tee -a gpurun_out/${tag}_var.log; tail -1 gpurun_out/${tag}.err | grep -v amdgpu.ids; }
runc GK_JIT_CHAIN=0
runc GK_JIT_CHAIN=1
runc GK_JIT_CHAIN=1 GK_JIT_LOOP_WEIGHT=1
runc GK_JIT_CHAIN=0
runc GK_JIT_CHAIN=1
GK_JIT_STRICT=1 timeout 600 python -m pytest tests -m gpu -x -q -k "referential" 2>&1 | tail -3 | tee gpurun_out/${tag}_pytest_gpu.log
