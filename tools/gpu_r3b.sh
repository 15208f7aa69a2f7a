#!/bin/bash
# GPU visit r3b: value-id accumulators (34 words per review): full gpu test suite, then the dominant kernel per geometry /
# register budget, phase clocks, kernel stats and SQ counters of the default build
set -u
tag=${1:-r3b}
mkdir -p gpurun_out; export TMPDIR=/tmp
GK_JIT_STRICT=1 timeout 1200 python -m pytest tests -m gpu -x -q 2>&1 | tail -5 > gpurun_out/${tag}_pytest_gpu.log
run() { env "$@" timeout 300 python bench.py --steps 40 --warmup 3 --no-cpu-baseline 2>gpurun_out/${tag}.err | python -c "
import sys, json
d = json.loads(sys.stdin.read().strip().splitlines()[-1])
print('$*  avg_kernel_ms %.4f ms_per_step %.4f frac %.4f lds %d pairs %d' % (d['roofline']['avg_kernel_ms'], d['ms_per_step'], d['roofline']['frac'], d['roofline']['lds_bytes_per_tile'], d['config']['violating_pairs_rank0']))" | tee -a gpurun_out/${tag}_var.log; }
run GK_X=0
run GK_PERSIST=2
run GK_JIT_WAVES=5
run GK_RPT=128
run GK_RPT=128 GK_JIT_WAVES=6
run GK_RPT=128 GK_JIT_WAVES=5
run GK_RPT=128 GK_JIT_WAVES=8
run GK_RPT=64
GK_KERNEL_PROF=1 timeout 300 python bench.py --steps 10 --warmup 2 --no-cpu-baseline 2>&1 >/dev/null | grep "gkgpu prof" | tail -1 | tee -a gpurun_out/${tag}_var.log
GK_RPT=128 GK_KERNEL_PROF=1 timeout 300 python bench.py --steps 10 --warmup 2 --no-cpu-baseline 2>&1 >/dev/null | grep "gkgpu prof" | tail -1 | tee -a gpurun_out/${tag}_var.log
(cd /tmp && timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/${tag}_stats -o stats -- python $GRAFT_REPO_ROOT/bench.py --steps 50 --warmup 5 --no-cpu-baseline > /dev/null 2> $GRAFT_REPO_ROOT/gpurun_out/${tag}_stats.err)
GK_BENCH_ARGS="" bash tools/gpu_pmc.sh ${tag}pmc > gpurun_out/${tag}_pmc.log 2>&1
tail -3 gpurun_out/${tag}_pytest_gpu.log
find gpurun_out/${tag}_stats -name '*kernel_stats.csv' | head -1 | xargs -r head -4
cat gpurun_out/${tag}_pmc.log | tail -30
