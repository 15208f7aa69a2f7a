#!/bin/bash
# quick A/B of kernel variants through environment knobs: tools/gpu_try.sh <tag> "ENV1=.. ENV2=.." "ENV.."
tag=$1; shift; mkdir -p gpurun_out
for envs in "$@"; do
  env $envs python bench.py --steps 50 --warmup 3 --no-cpu-baseline 2>gpurun_out/${tag}_try.err | python -c "
import sys, json
d = json.loads(sys.stdin.read().strip().splitlines()[-1])
print('[$envs] avg_kernel_ms %.4f ms_per_step %.4f frac %.4f lds %d pairs %d' % (d['roofline']['avg_kernel_ms'], d['ms_per_step'], d['roofline']['frac'], d['roofline']['lds_bytes_per_tile'], d['config']['violating_pairs_rank0']))" | tee -a gpurun_out/${tag}_try.log
  tail -2 gpurun_out/${tag}_try.err | grep -v amdgpu.ids
done
