#!/bin/bash
# GPU visit ab: ahead-requests without drains -- parity subset, A/B, clock profile
mkdir -p gpurun_out
GK_JIT_STRICT=1 timeout 900 python -m pytest tests -m gpu -x -q -k "row_group or synthetic_parity or edge_cases or config1_demo" 2>&1 | tail -3 | tee gpurun_out/r2ab_pytest.log
bash tools/gpu_variants.sh r2ab - "-,GK_JIT_PREFETCH=2" "-,GK_PERSIST=0 GK_JIT_PREFETCH=2" "-,GK_JIT_PREFETCH=2 GK_LOOP_UNROLL=2" "-,GK_JIT_PREFETCH=2 GK_LOOP_UNROLL=4" "-,GK_JIT_PREFETCH=2 GK_RPT=128" "-,GK_JIT_PREFETCH=2"
for pf in 3 2; do GK_JIT_PREFETCH=$pf GK_KERNEL_PROF=1 python bench.py --steps 5 --warmup 1 --no-cpu-baseline 2>&1 | grep "gkgpu prof" | tail -1 | tee -a gpurun_out/r2ab_prof.log; done
