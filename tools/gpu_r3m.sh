#!/bin/bash
# Round-3 visit m: plain class switch as the default (GK_JIT_HOT=0), list entries in registers (GK_LIST_REGS), full-unroll bound
set -u
tag=${1:-r3m}
mkdir -p gpurun_out; export TMPDIR=/tmp
run() { env "$@" python bench.py --steps 100 --warmup 5 --lean 2>gpurun_out/${tag}.err | python -c "
import sys, json
d = json.loads(sys.stdin.read().strip().splitlines()[-1])
print('$*  avg_kernel_ms %.4f ms_per_step %.4f frac %.4f pairs %d' % (d['roofline']['avg_kernel_ms'], d['ms_per_step'], d['roofline']['frac'], d['config']['violating_pairs_rank0']))" | tee -a gpurun_out/${tag}_var.log; tail -1 gpurun_out/${tag}.err | grep -v amdgpu.ids; }
run GK_X=0
run GK_JIT_DEFINES=GK_LIST_REGS
run GK_JIT_HOT=1
run GK_UNROLL_MAX=16
run GK_JIT_DEFINES=GK_LIST_REGS GK_UNROLL_MAX=16
run GK_JIT_HOT=4
run GK_X=0
run GK_JIT_DEFINES=GK_LIST_REGS
GK_KERNEL_PROF=1 python bench.py --steps 20 --warmup 3 --lean 2>&1 >/dev/null | grep "gkgpu prof" | tail -1 | tee -a gpurun_out/${tag}_var.log
GK_JIT_DEFINES=GK_LIST_REGS GK_KERNEL_PROF=1 python bench.py --steps 20 --warmup 3 --lean 2>&1 >/dev/null | grep "gkgpu prof" | tail -1 | tee -a gpurun_out/${tag}_var.log
