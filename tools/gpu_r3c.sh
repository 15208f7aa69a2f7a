#!/bin/bash
# GPU visit r3c: A/B of the flat (branch-free) row bodies and the dynamic group order on configs[2]; background-JIT test;
# the streaming configs[4] record; native admission storm through the per-table streams
set -u
tag=${1:-r3c}
mkdir -p gpurun_out; export TMPDIR=/tmp
run() { env "$@" timeout 300 python bench.py --steps 40 --warmup 3 --no-cpu-baseline 2>gpurun_out/${tag}.err | python -c "
import sys, json
d = json.loads(sys.stdin.read().strip().splitlines()[-1])
print('$*  avg_kernel_ms %.4f ms_per_step %.4f frac %.4f lds %d pairs %d' % (d['roofline']['avg_kernel_ms'], d['ms_per_step'], d['roofline']['frac'], d['roofline']['lds_bytes_per_tile'], d['config']['violating_pairs_rank0']))" | tee -a gpurun_out/${tag}_var.log; }
run GK_X=0
run GK_JIT_FLAT=0
run GK_DYN_GROUPS=0
run GK_JIT_FLAT=0 GK_DYN_GROUPS=0
run GK_NULL_STREAM=1
run GK_JIT_HOT=0
run GK_JIT_HOT=8
GK_KERNEL_PROF=1 timeout 300 python bench.py --steps 10 --warmup 2 --no-cpu-baseline 2>&1 >/dev/null | grep "gkgpu prof" | tail -1 | tee -a gpurun_out/${tag}_var.log
timeout 600 python -m pytest tests/test_jit_background.py tests/test_batcher.py tests/test_sweep_dist.py tests/test_resident.py -m gpu -x -q 2>&1 | tail -15 > gpurun_out/${tag}_pytest_gpu.log
timeout 600 python bench.py --config 4 --streaming --offered 0 --stream-batches 12 > gpurun_out/${tag}_stream_closed.json 2> gpurun_out/${tag}_stream_closed.err
timeout 600 python bench.py --config 4 --streaming --offered 1000000 --stream-batches 12 > gpurun_out/${tag}_stream_1M.json 2> gpurun_out/${tag}_stream_1M.err
timeout 600 python tools/latency_probe.py > gpurun_out/${tag}_latency.json 2> gpurun_out/${tag}_latency.err
cat gpurun_out/${tag}_pytest_gpu.log
head -c 2500 gpurun_out/${tag}_stream_closed.json; echo; tail -3 gpurun_out/${tag}_stream_closed.err
head -c 1500 gpurun_out/${tag}_stream_1M.json; echo
tail -c 1800 gpurun_out/${tag}_latency.json; echo; tail -3 gpurun_out/${tag}_latency.err
