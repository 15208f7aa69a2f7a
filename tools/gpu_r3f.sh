#!/bin/bash
# GPU visit r3f: dynamic group order with the ticket drawn in phase 2; corpus sweep per row-group size (plan-specialised kernel
# awaited); background-JIT test
set -u
tag=${1:-r3f}
mkdir -p gpurun_out; export TMPDIR=/tmp
run() { env "$@" timeout 300 python bench.py --steps 40 --warmup 3 --no-cpu-baseline 2>gpurun_out/${tag}.err | python -c "
import sys, json
d = json.loads(sys.stdin.read().strip().splitlines()[-1])
print('$*  avg_kernel_ms %.4f ms_per_step %.4f frac %.4f flatten_s %.3f' % (d['roofline']['avg_kernel_ms'], d['ms_per_step'], d['roofline']['frac'], d['end_to_end']['flatten_s']))" | tee -a gpurun_out/${tag}_var.log; }
run GK_X=0
run GK_DYN_GROUPS=1
run GK_X=1
run GK_DYN_GROUPS=1 GK_X=1
run GK_PERSIST=0
GK_DYN_GROUPS=1 GK_KERNEL_PROF=1 timeout 300 python bench.py --steps 10 --warmup 2 --no-cpu-baseline 2>&1 >/dev/null | grep "gkgpu prof" | tail -1 | tee -a gpurun_out/${tag}_var.log
c4() { env "$@" timeout 300 python bench.py --config 4 --steps 20 --warmup 3 --no-cpu-baseline 2>gpurun_out/${tag}.err | python -c "
import sys, json
d = json.loads(sys.stdin.read().strip().splitlines()[-1])
print('config4 $*  ms_per_step %.4f value %.3g kernel_ms(sum of groups) %.4f lds %d algo_bytes %d' % (d['ms_per_step'], d['value'], d['roofline']['avg_kernel_ms'], d['roofline']['lds_bytes_per_tile'], d['roofline']['algo_bytes_per_launch']))" | tee -a gpurun_out/${tag}_var.log; }
c4 GK_X=0
c4 GK_RPT=128
c4 GK_RPT=64
c4 GK_NULL_STREAM=1
timeout 600 python -m pytest tests/test_jit_background.py tests/test_kernel_emu.py tests/test_parity.py -m gpu -x -q -k "jit or synthetic or row_group or edge" 2>&1 | tail -15 > gpurun_out/${tag}_pytest_gpu.log
cat gpurun_out/${tag}_pytest_gpu.log | tail -8
