#!/bin/bash
# GPU visit ag: what phase 1 waits for -- the walk without loads (GK_DBG_PHASE 32), and wave 0's vmcnt waits (GK_PROF_WAIT)
mkdir -p gpurun_out
export GK_JIT_PREFETCH=1
run() { env "$@" GK_KERNEL_PROF=1 python bench.py --steps 20 --warmup 2 --no-cpu-baseline 2>gpurun_out/r2ag.err | python -c "
import sys, json
d = json.loads(sys.stdin.read().strip().splitlines()[-1])
print('$*  avg_kernel_ms %.4f' % d['roofline']['avg_kernel_ms'])" | tee -a gpurun_out/r2ag.log; grep "gkgpu prof" gpurun_out/r2ag.err | tail -2 | tee -a gpurun_out/r2ag.log; }
run GK_DBG_PHASE=33
run GK_DBG_PHASE=32
run GK_DBG_PHASE=1 GK_JIT_DEFINES=GK_PROF_WAIT
run GK_DBG_PHASE=0 GK_JIT_DEFINES=GK_PROF_WAIT
run GK_DBG_PHASE=0 GK_JIT_DEFINES=GK_PROF_WAIT GK_JIT_PREFETCH=2
run GK_DBG_PHASE=0 GK_JIT_DEFINES=GK_PROF_WAIT GK_JIT_PREFETCH=3
