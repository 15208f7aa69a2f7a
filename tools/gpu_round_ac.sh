#!/bin/bash
# GPU visit ac: phase isolation of the persistent kernel (GK_DBG_PHASE bits: 1 rows loaded but not evaluated, 16 no chunks,
# 2 no formulas, 64 no outputs) with the in-kernel clock profile, persistent and one-workgroup-per-group
mkdir -p gpurun_out
export GK_JIT_PREFETCH=2
for ph in 0 1 16 2 64 3 18; do
  GK_DBG_PHASE=$ph GK_KERNEL_PROF=1 python bench.py --steps 20 --warmup 2 --no-cpu-baseline 2>gpurun_out/r2ac.err | python -c "
import sys, json
d = json.loads(sys.stdin.read().strip().splitlines()[-1])
print('GK_DBG_PHASE=$ph avg_kernel_ms %.4f' % d['roofline']['avg_kernel_ms'])" | tee -a gpurun_out/r2ac_phases.log
  grep "gkgpu prof" gpurun_out/r2ac.err | tail -1 | tee -a gpurun_out/r2ac_phases.log
done
GK_PERSIST=0 GK_KERNEL_PROF=1 python bench.py --steps 20 --warmup 2 --no-cpu-baseline 2>&1 | grep "gkgpu prof" | tail -1 | tee -a gpurun_out/r2ac_phases.log
