#!/bin/bash
# GPU visit r3h: list buffers sized by the table (static LDS 8.4 -> 5.4 KB): FOUR 8-wave groups per CU at 64 VGPRs against three at 80
set -u
tag=${1:-r3h}
mkdir -p gpurun_out; export TMPDIR=/tmp
run() { env "$@" timeout 300 python bench.py --steps 40 --warmup 3 --no-cpu-baseline 2>gpurun_out/${tag}.err | python -c "
import sys, json
d = json.loads(sys.stdin.read().strip().splitlines()[-1])
print('$*  avg_kernel_ms %.4f ms_per_step %.4f frac %.4f pairs %d' % (d['roofline']['avg_kernel_ms'], d['ms_per_step'], d['roofline']['frac'], d['config']['violating_pairs_rank0']))" | tee -a gpurun_out/${tag}_var.log; }
run GK_X=0
run GK_JIT_FULL_LISTS=1
run GK_X=1
run GK_JIT_FULL_LISTS=1 GK_X=1
run GK_JIT_WAVES=7
run GK_PERSIST=3
GK_KERNEL_PROF=1 timeout 300 python bench.py --steps 10 --warmup 2 --no-cpu-baseline 2>&1 >/dev/null | grep "gkgpu prof" | tail -1 | tee -a gpurun_out/${tag}_var.log
timeout 600 python -m pytest tests -m gpu -x -q -k "synthetic_parity or row_group or edge_cases or structural_fuzz or kernel or bench_legs" 2>&1 | tail -4 | tee gpurun_out/${tag}_pytest_gpu.log
