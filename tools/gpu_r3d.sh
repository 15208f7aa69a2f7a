#!/bin/bash
# GPU visit r3d: phase budget of the dominant kernel at three workgroups per CU (time and SQ instruction counts per phase toggle),
# admission storm after the pinned-pool fix, streaming configs[4] with persistent flatteners, host ingest scaling, corpus sweep
set -u
tag=${1:-r3d}
mkdir -p gpurun_out; export TMPDIR=/tmp
run() { env "$@" timeout 300 python bench.py --steps 40 --warmup 3 --no-cpu-baseline 2>gpurun_out/${tag}.err | python -c "
import sys, json
d = json.loads(sys.stdin.read().strip().splitlines()[-1])
print('$*  avg_kernel_ms %.4f ms_per_step %.4f frac %.4f flatten_s %.3f h2d_s %.3f threads %d' % (d['roofline']['avg_kernel_ms'], d['ms_per_step'], d['roofline']['frac'], d['end_to_end']['flatten_s'], d['end_to_end']['h2d_s'], d['end_to_end']['host_threads']))" | tee -a gpurun_out/${tag}_var.log; }
run GK_X=0
for ph in 1 2 16 64 3 18; do run GK_DBG_PHASE=$ph; done
run GK_HOST_THREADS=64
run GK_HOST_THREADS=128
run GK_HOST_SPAWN=1
pmc() { name=$1; shift; env "$@" rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY --kernel-trace --output-format csv -d gpurun_out/${tag}pmc_$name -o sq -- python bench.py --steps 5 --warmup 1 --no-cpu-baseline > /dev/null 2> gpurun_out/${tag}pmc_$name.err
  f=$(find gpurun_out/${tag}pmc_$name -name '*counter_collection.csv' | head -1)
  [ -n "$f" ] && python - "$f" "$name" <<'PY' | tee -a gpurun_out/${tag}_pmc.log
import csv, sys, collections
acc = collections.defaultdict(float); n = collections.Counter()
for r in csv.DictReader(open(sys.argv[1])):
    if 'tiles' not in r.get('Kernel_Name', ''): continue
    acc[r['Counter_Name']] += float(r['Counter_Value']); n[r['Counter_Name']] += 1
print(sys.argv[2], ' '.join('%s=%.0f' % (c, v / n[c]) for c, v in sorted(acc.items())))
PY
}
pmc phase0 GK_X=0
pmc phase2 GK_DBG_PHASE=2
pmc phase18 GK_DBG_PHASE=18
pmc phase1 GK_DBG_PHASE=1
timeout 600 python -m pytest tests/test_jit_background.py tests/test_spool.py -m gpu -x -q 2>&1 | tail -4 | tee gpurun_out/${tag}_pytest_gpu.log
timeout 600 python tools/latency_probe.py > gpurun_out/${tag}_latency.json 2> gpurun_out/${tag}_latency.err
GK_NULL_STREAM=1 timeout 600 python tools/latency_probe.py > gpurun_out/${tag}_latency_nullstream.json 2> gpurun_out/${tag}_latency_nullstream.err
timeout 600 python bench.py --config 4 --streaming --offered 0 --stream-batches 16 > gpurun_out/${tag}_stream_closed.json 2> gpurun_out/${tag}_stream_closed.err
timeout 600 python bench.py --config 4 --steps 20 --warmup 3 --no-cpu-baseline > gpurun_out/${tag}_bench_config4.json 2> gpurun_out/${tag}_bench_config4.err
python - <<'PY'
import json
for f in ('latency', 'latency_nullstream'):
    try:
        d = json.load(open('gpurun_out/r3d_%s.json' % f))
        for r in d.get('native', {}).get('runs', d.get('runs', [])):
            if r.get('threads') in (1, 64, 256): print(f, {k: (round(v, 1) if isinstance(v, float) else v) for k, v in r.items() if k in ('threads', 'p50_us', 'p99_us', 'reviews_per_s', 'mean_batch', 'window_us', 'workers', 'max_batch')})
    except Exception as e: print(f, 'ERR', e)
for f in ('stream_closed', 'bench_config4'):
    try:
        d = json.loads(open('gpurun_out/r3d_%s.json' % f).read().strip().splitlines()[-1])
        print(f, 'value %.3g ms_per_step %.3f' % (d['value'], d['ms_per_step']), json.dumps(d.get('stream', d.get('roofline')))[:900])
    except Exception as e: print(f, 'ERR', e)
PY
