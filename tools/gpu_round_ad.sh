#!/bin/bash
# GPU visit ad: chunks in flight per wave (GK_JIT_PREFETCH) with and without row evaluation (GK_DBG_PHASE=1)
mkdir -p gpurun_out
for pf in 1 2 3 4 5; do for ph in 0 1; do
  GK_JIT_PREFETCH=$pf GK_DBG_PHASE=$ph python bench.py --steps 30 --warmup 2 --no-cpu-baseline 2>gpurun_out/r2ad.err | python -c "
import sys, json
d = json.loads(sys.stdin.read().strip().splitlines()[-1])
print('GK_JIT_PREFETCH=$pf GK_DBG_PHASE=$ph avg_kernel_ms %.4f' % d['roofline']['avg_kernel_ms'])" | tee -a gpurun_out/r2ad_prefetch.log
done; done
for pf in 2 4; do GK_RPT=128 GK_JIT_PREFETCH=$pf python bench.py --steps 30 --warmup 2 --no-cpu-baseline 2>gpurun_out/r2ad.err | python -c "
import sys, json
d = json.loads(sys.stdin.read().strip().splitlines()[-1])
print('GK_RPT=128 GK_JIT_PREFETCH=$pf avg_kernel_ms %.4f' % d['roofline']['avg_kernel_ms'])" | tee -a gpurun_out/r2ad_prefetch.log; done
