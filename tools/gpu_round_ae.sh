#!/bin/bash
# GPU visit ae: late start of every other workgroup (GK_STAGGER x 256 clocks; mode 0: odd positions, 1: upper half of each XCD's workgroups)
mkdir -p gpurun_out
run() { env "$@" python bench.py --steps 30 --warmup 2 --no-cpu-baseline 2>gpurun_out/r2ae.err | python -c "
import sys, json
d = json.loads(sys.stdin.read().strip().splitlines()[-1])
print('$*  avg_kernel_ms %.4f ms_per_step %.4f pairs %d' % (d['roofline']['avg_kernel_ms'], d['ms_per_step'], d['config']['violating_pairs_rank0']))" | tee -a gpurun_out/r2ae_stagger.log; }
for st in 0 48 96 144; do for mode in 0 1; do run GK_JIT_PREFETCH=1 GK_STAGGER=$st GK_STAGGER_MODE=$mode; [ $st = 0 ] && break; done; done
run GK_JIT_PREFETCH=2 GK_STAGGER=96 GK_STAGGER_MODE=0
run GK_JIT_PREFETCH=2 GK_STAGGER=96 GK_STAGGER_MODE=1
run GK_JIT_PREFETCH=1 GK_STAGGER=96 GK_STAGGER_MODE=1 GK_PERSIST=0
GK_JIT_PREFETCH=1 GK_STAGGER=96 GK_STAGGER_MODE=1 GK_KERNEL_PROF=1 python bench.py --steps 5 --warmup 1 --no-cpu-baseline 2>&1 | grep "gkgpu prof" | tail -1 | tee -a gpurun_out/r2ae_stagger.log
