#!/bin/bash
# phase-isolation timings of the dominant kernel (GK_DBG_PHASE bits: 1 skip predicate evaluation, 2 skip formulas, 8 zero the results,
# 16 skip phase 1 entirely, 64 skip the bitmap stores, 128 skip the accumulator clearing; results are wrong with any bit set)
tag=${1:-ph}; shift; mkdir -p gpurun_out
for ph in ${@:-0 1 2 3 16 18 64 66 82}; do
  GK_DBG_PHASE=$ph python bench.py --steps 30 --warmup 3 --no-cpu-baseline 2>/dev/null | python -c "
import sys, json
d = json.loads(sys.stdin.read().strip().splitlines()[-1])
print('GK_DBG_PHASE=$ph avg_kernel_ms %.4f' % d['roofline']['avg_kernel_ms'])" | tee -a gpurun_out/${tag}_phases.log
done
