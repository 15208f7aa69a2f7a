#!/bin/bash
# GPU visit ah: walk by class vs class switch in the chunk loop
mkdir -p gpurun_out
GK_JIT_STRICT=1 timeout 900 python -m pytest tests -m gpu -x -q -k "row_group or synthetic_parity or edge_cases or config1_demo" 2>&1 | tail -3 | tee gpurun_out/r2ah_pytest.log
run() { env "$@" python bench.py --steps 30 --warmup 2 --no-cpu-baseline 2>gpurun_out/r2ah.err | python -c "
import sys, json
d = json.loads(sys.stdin.read().strip().splitlines()[-1])
print('$*  avg_kernel_ms %.4f ms_per_step %.4f frac %.4f pairs %d' % (d['roofline']['avg_kernel_ms'], d['ms_per_step'], d['roofline']['frac'], d['config']['violating_pairs_rank0']))" | tee -a gpurun_out/r2ah_var.log; }
run GK_X=1
run GK_JIT_DEFINES=GK_NO_CLASS_WALK
run GK_X=2
run GK_DBG_PHASE=1
run GK_PERSIST=0
GK_KERNEL_PROF=1 python bench.py --steps 5 --warmup 1 --no-cpu-baseline 2>&1 | grep "gkgpu prof" | tail -1 | tee -a gpurun_out/r2ah_var.log
GK_DBG_PHASE=1 GK_KERNEL_PROF=1 python bench.py --steps 5 --warmup 1 --no-cpu-baseline 2>&1 | grep "gkgpu prof" | tail -1 | tee -a gpurun_out/r2ah_var.log
