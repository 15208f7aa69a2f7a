#!/bin/bash
# phase isolation per row-group geometry: tools/gpu_phases2.sh <tag> "<rpt list>" "<phase list>"
tag=$1; mkdir -p gpurun_out
for rpt in $2; do for ph in $3; do
  GK_RPT=$rpt GK_DBG_PHASE=$ph python bench.py --steps 30 --warmup 3 --no-cpu-baseline 2>/dev/null | python -c "
import sys, json
d = json.loads(sys.stdin.read().strip().splitlines()[-1])
print('GK_RPT=$rpt GK_DBG_PHASE=$ph avg_kernel_ms %.4f' % d['roofline']['avg_kernel_ms'])" | tee -a gpurun_out/${tag}_phases.log
done; done
