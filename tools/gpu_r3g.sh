#!/bin/bash
# GPU visit r3g: runs of one predicate class per wave (+ list entry read one chunk ahead) against one dispatch per chunk;
# corpus sweep with the plan-aware row-group size; storm; quick parity on the GPU
set -u
tag=${1:-r3g}
mkdir -p gpurun_out; export TMPDIR=/tmp
run() { env "$@" timeout 300 python bench.py --steps 40 --warmup 3 --no-cpu-baseline 2>gpurun_out/${tag}.err | python -c "
import sys, json
d = json.loads(sys.stdin.read().strip().splitlines()[-1])
print('$*  avg_kernel_ms %.4f ms_per_step %.4f frac %.4f flatten_s %.3f pairs %d' % (d['roofline']['avg_kernel_ms'], d['ms_per_step'], d['roofline']['frac'], d['end_to_end']['flatten_s'], d['config']['violating_pairs_rank0']))" | tee -a gpurun_out/${tag}_var.log; }
run GK_X=0
run GK_JIT_RUNS=0
run GK_X=1
run GK_JIT_RUNS=0 GK_X=1
run GK_JIT_HOT=0
run GK_JIT_HOT=8
GK_KERNEL_PROF=1 timeout 300 python bench.py --steps 10 --warmup 2 --no-cpu-baseline 2>&1 >/dev/null | grep "gkgpu prof" | tail -1 | tee -a gpurun_out/${tag}_var.log
c4() { env "$@" timeout 300 python bench.py --config 4 --steps 20 --warmup 3 --no-cpu-baseline 2>gpurun_out/${tag}.err | python -c "
import sys, json
d = json.loads(sys.stdin.read().strip().splitlines()[-1])
print('config4 $*  ms_per_step %.4f value %.3g kernel_ms(sum of groups) %.4f lds %d algo_bytes %d' % (d['ms_per_step'], d['value'], d['roofline']['avg_kernel_ms'], d['roofline']['lds_bytes_per_tile'], d['roofline']['algo_bytes_per_launch']))" | tee -a gpurun_out/${tag}_var.log; }
c4 GK_X=0
c4 GK_JIT_RUNS=0
timeout 900 python -m pytest tests -m gpu -x -q -k "jit_background or synthetic_parity or row_group or edge_cases or structural_fuzz or library_patterns_one_plan or corpus or kernel or spool or batcher" 2>&1 | tail -12 > gpurun_out/${tag}_pytest_gpu.log
cat gpurun_out/${tag}_pytest_gpu.log | tail -6
