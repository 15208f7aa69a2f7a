#!/bin/bash
# Round-3 visit l: the remaining tuning aids re-measured on the round's final kernel (hot-class chain length, full-unroll bound of
# the formula loops, late start of every other workgroup), one box
set -u
tag=${1:-r3l}
mkdir -p gpurun_out; export TMPDIR=/tmp
run() { env "$@" python bench.py --steps 100 --warmup 5 --lean 2>gpurun_out/${tag}.err | python -c "
import sys, json
d = json.loads(sys.stdin.read().strip().splitlines()[-1])
print('$*  avg_kernel_ms %.4f ms_per_step %.4f frac %.4f pairs %d' % (d['roofline']['avg_kernel_ms'], d['ms_per_step'], d['roofline']['frac'], d['config']['violating_pairs_rank0']))" | tee -a gpurun_out/${tag}_var.log; tail -1 gpurun_out/${tag}.err | grep -v amdgpu.ids; }
run GK_X=0
run GK_JIT_HOT=0
run GK_JIT_HOT=2
run GK_JIT_HOT=6
run GK_UNROLL_MAX=2
run GK_UNROLL_MAX=8
run GK_UNROLL_MAX=16
run GK_STAGGER=24
run GK_STAGGER=48
run GK_X=0
