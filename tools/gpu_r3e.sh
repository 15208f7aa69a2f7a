#!/bin/bash
# GPU visit r3e: loads in flight per wave (GK_JIT_PREFETCH) and formula loop unrolling at three workgroups per CU; admission
# storm with concurrent evaluations + single-part tables; corpus sweep with overlapping plan groups, per row-group size
set -u
tag=${1:-r3e}
mkdir -p gpurun_out; export TMPDIR=/tmp
run() { env "$@" timeout 300 python bench.py --steps 40 --warmup 3 --no-cpu-baseline 2>gpurun_out/${tag}.err | python -c "
import sys, json
d = json.loads(sys.stdin.read().strip().splitlines()[-1])
print('$*  avg_kernel_ms %.4f ms_per_step %.4f frac %.4f flatten_s %.3f' % (d['roofline']['avg_kernel_ms'], d['ms_per_step'], d['roofline']['frac'], d['end_to_end']['flatten_s']))" | tee -a gpurun_out/${tag}_var.log; }
run GK_X=0
run GK_JIT_PREFETCH=2
run GK_JIT_PREFETCH=3
run GK_UNROLL_MAX=12
run GK_LOOP_UNROLL=2
run GK_LOOP_UNROLL=4
run GK_JIT_PREFETCH=2 GK_LOOP_UNROLL=2
run GK_RPT=128 GK_JIT_PREFETCH=2
run GK_X=1
c4() { env "$@" timeout 300 python bench.py --config 4 --steps 20 --warmup 3 --no-cpu-baseline 2>gpurun_out/${tag}.err | python -c "
import sys, json
d = json.loads(sys.stdin.read().strip().splitlines()[-1])
print('config4 $*  ms_per_step %.4f value %.3g kernel_ms(sum of groups) %.4f lds %d algo_bytes %d' % (d['ms_per_step'], d['value'], d['roofline']['avg_kernel_ms'], d['roofline']['lds_bytes_per_tile'], d['roofline']['algo_bytes_per_launch']))" | tee -a gpurun_out/${tag}_var.log; }
c4 GK_X=0
c4 GK_RPT=128
c4 GK_RPT=64
timeout 600 python -m pytest tests/test_jit_background.py tests/test_spool.py tests/test_batcher.py -m gpu -x -q 2>&1 | tail -40 > gpurun_out/${tag}_pytest_gpu.log
timeout 600 python tools/latency_probe.py > gpurun_out/${tag}_latency.json 2> gpurun_out/${tag}_latency.err
python - <<'PY'
import json
try:
    d = json.load(open('gpurun_out/r3e_latency.json'))
    for r in d.get('native', {}).get('runs', d.get('runs', [])):
        if r.get('threads') in (1, 64, 256): print('latency', {k: (round(v, 1) if isinstance(v, float) else v) for k, v in r.items() if k in ('threads', 'p50_us', 'p99_us', 'reviews_per_s', 'mean_batch', 'window_us', 'workers', 'max_batch')})
except Exception as e: print('latency ERR', e)
PY
cat gpurun_out/${tag}_pytest_gpu.log | tail -25
