#!/bin/bash
# A/B of kernel_body.inc variants in ONE GPU call (same box, same clocks): tools/gpu_variants.sh <tag> <file|-> ["ENV=.."]...
tag=$1; shift; mkdir -p gpurun_out
for spec in "$@"; do
  f=${spec%%,*}; envs=""; [ "$spec" != "$f" ] && envs=${spec#*,}
  if [ "$f" != "-" ]; then export GK_JIT_BODY_FILE=$PWD/$f; else unset GK_JIT_BODY_FILE; fi
  env $envs python bench.py --steps 50 --warmup 3 --no-cpu-baseline 2>gpurun_out/${tag}_var.err | python -c "
import sys, json
d = json.loads(sys.stdin.read().strip().splitlines()[-1])
print('[$spec] avg_kernel_ms %.4f ms_per_step %.4f frac %.4f pairs %d' % (d['roofline']['avg_kernel_ms'], d['ms_per_step'], d['roofline']['frac'], d['config']['violating_pairs_rank0']))" | tee -a gpurun_out/${tag}_var.log
  tail -2 gpurun_out/${tag}_var.err | grep -v amdgpu.ids
done
