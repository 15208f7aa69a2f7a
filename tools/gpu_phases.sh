#!/bin/bash
# phase-isolation timings of the dominant kernel (GK_DBG_PHASE bits: 1 skip predicates, 2 skip formulas, 4 skip count atomics, 8 suppress outputs)
tag=${1:-ph}; mkdir -p gpurun_out
for ph in 0 1 2 3 4 8 12; do
  GK_DBG_PHASE=$ph python bench.py --steps 30 --warmup 3 --no-cpu-baseline 2>/dev/null | python -c "
import sys, json
d = json.loads(sys.stdin.read().strip().splitlines()[-1])
print('GK_DBG_PHASE=$ph avg_kernel_ms %.4f' % d['roofline']['avg_kernel_ms'])" | tee -a gpurun_out/${tag}_phases.log
done
