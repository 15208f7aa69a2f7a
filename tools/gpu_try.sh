#!/bin/bash
# quick A/B of kernel variants through environment knobs: tools/gpu_try.sh <tag> "ENV1=.. ENV2=.." "ENV.."
tag=$1; shift; mkdir -p gpurun_out
for envs in "$@"; do
  env $envs python bench.py --steps 30 --warmup 3 --no-cpu-baseline 2>/dev/null | python -c "
import sys, json
d = json.loads(sys.stdin.read().strip().splitlines()[-1])
print('[$envs] avg_kernel_ms %.4f ms_per_step %.4f' % (d['roofline']['avg_kernel_ms'], d['ms_per_step']))" | tee -a gpurun_out/${tag}_try.log
done
