#!/bin/bash
# GPU visit r3a: occupancy sensitivity of the dominant kernel (configs[2]) -- workgroups per CU (GK_PERSIST), row-group
# geometry (GK_RPT) and waves per group (GK_JIT_BLOCK); phase clocks of the default build
set -u
tag=${1:-r3a}
mkdir -p gpurun_out; export TMPDIR=/tmp
run() { env "$@" timeout 300 python bench.py --steps 40 --warmup 3 --no-cpu-baseline 2>gpurun_out/${tag}.err | python -c "
import sys, json
d = json.loads(sys.stdin.read().strip().splitlines()[-1])
print('$*  avg_kernel_ms %.4f ms_per_step %.4f frac %.4f lds %d pairs %d' % (d['roofline']['avg_kernel_ms'], d['ms_per_step'], d['roofline']['frac'], d['roofline']['lds_bytes_per_tile'], d['config']['violating_pairs_rank0']))" | tee -a gpurun_out/${tag}_var.log; }
run GK_X=0
run GK_PERSIST=1
run GK_RPT=128
run GK_RPT=128 GK_PERSIST=2
run GK_RPT=128 GK_JIT_BLOCK=512
run GK_RPT=64
run GK_RPT=64 GK_PERSIST=4
run GK_JIT_WAVES=5
GK_KERNEL_PROF=1 timeout 300 python bench.py --steps 10 --warmup 2 --no-cpu-baseline 2>&1 >/dev/null | grep "gkgpu prof" | tail -2 | tee -a gpurun_out/${tag}_var.log
GK_RPT=128 GK_KERNEL_PROF=1 timeout 300 python bench.py --steps 10 --warmup 2 --no-cpu-baseline 2>&1 >/dev/null | grep "gkgpu prof" | tail -2 | tee -a gpurun_out/${tag}_var.log
