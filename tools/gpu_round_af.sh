#!/bin/bash
# GPU visit af: word-wise string predicates -- parity subset, prefetch depth, clock profile
mkdir -p gpurun_out
GK_JIT_STRICT=1 timeout 900 python -m pytest tests -m gpu -x -q -k "row_group or synthetic_parity or edge_cases or config1_demo or regex or fuzz or corpus" 2>&1 | tail -3 | tee gpurun_out/r2af_pytest.log
run() { env "$@" python bench.py --steps 30 --warmup 2 --no-cpu-baseline 2>gpurun_out/r2af.err | python -c "
import sys, json
d = json.loads(sys.stdin.read().strip().splitlines()[-1])
print('$*  avg_kernel_ms %.4f ms_per_step %.4f frac %.4f pairs %d' % (d['roofline']['avg_kernel_ms'], d['ms_per_step'], d['roofline']['frac'], d['config']['violating_pairs_rank0']))" | tee -a gpurun_out/r2af_var.log; }
run GK_JIT_PREFETCH=1
run GK_JIT_PREFETCH=2
run GK_JIT_PREFETCH=3
run GK_JIT_PREFETCH=1 GK_PERSIST=0
run GK_JIT_PREFETCH=2 GK_DBG_PHASE=1
for pf in 1 2; do GK_JIT_PREFETCH=$pf GK_KERNEL_PROF=1 python bench.py --steps 5 --warmup 1 --no-cpu-baseline 2>&1 | grep "gkgpu prof" | tail -1 | tee -a gpurun_out/r2af_var.log; done
