#!/bin/bash
# Round-3 records: the whole gpu test suite, smoke, the default bench line (configs[2]) with rocprofv3 kernel stats and PMC passes
# of the SAME build (SQ, FETCH_SIZE, WRITE_SIZE in separate passes, --kernel-trace only), configs[1], the corpus sweep, the
# streaming configs[4] records (closed loop and 1 M reviews/s offered), the native admission storm
set -u
tag=${1:-r3z}
mkdir -p gpurun_out; export TMPDIR=/tmp
GK_JIT_STRICT=1 timeout 1500 python -m pytest tests -m gpu -x -q 2>&1 | tail -6 > gpurun_out/${tag}_pytest_gpu.log
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" > gpurun_out/${tag}_smoke.log 2>&1
timeout 900 python bench.py > gpurun_out/${tag}_bench_config2.json 2> gpurun_out/${tag}_bench_config2.err
(cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/${tag}_stats -o stats -- python $GRAFT_REPO_ROOT/bench.py --steps 50 --warmup 5 --no-cpu-baseline > /dev/null 2> $GRAFT_REPO_ROOT/gpurun_out/${tag}_stats.err)
run_pmc() { name=$1; shift; rocprofv3 --pmc "$@" --kernel-trace --output-format csv -d gpurun_out/${tag}pmc_$name -o $name -- python bench.py --steps 5 --warmup 1 --no-cpu-baseline > /dev/null 2> gpurun_out/${tag}pmc_$name.err; }
run_pmc sq SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY
run_pmc fetch FETCH_SIZE
run_pmc write WRITE_SIZE
for d in sq fetch write; do
  f=$(find gpurun_out/${tag}pmc_$d -name '*counter_collection.csv' | head -1)
  [ -n "$f" ] && python - "$f" <<'PY' >> gpurun_out/${tag}_pmc.log
import csv, sys, collections
acc = collections.defaultdict(lambda: collections.defaultdict(float)); n = collections.Counter()
for r in csv.DictReader(open(sys.argv[1])):
    k = r.get('Kernel_Name', '')
    if 'tiles' not in k: continue
    acc[k][r['Counter_Name']] += float(r['Counter_Value']); n[(k, r['Counter_Name'])] += 1
for k in acc:
    for c, v in sorted(acc[k].items()):
        print('%s %s per_dispatch=%.1f dispatches=%d' % (k[:20], c, v / n[(k, c)], n[(k, c)]))
PY
done
timeout 600 python bench.py --config 1 --no-cpu-baseline > gpurun_out/${tag}_bench_config1.json 2> gpurun_out/${tag}_bench_config1.err
timeout 600 python bench.py --config 4 --steps 50 --warmup 5 --no-cpu-baseline > gpurun_out/${tag}_bench_config4.json 2> gpurun_out/${tag}_bench_config4.err
timeout 600 python bench.py --config 4 --streaming --offered 0 --stream-batches 16 > gpurun_out/${tag}_stream_closed.json 2> gpurun_out/${tag}_stream_closed.err
timeout 600 python bench.py --config 4 --streaming --offered 1000000 --stream-batches 16 > gpurun_out/${tag}_stream_1M.json 2> gpurun_out/${tag}_stream_1M.err
timeout 600 python tools/latency_probe.py > gpurun_out/${tag}_latency.json 2> gpurun_out/${tag}_latency.err
GK_FORCE_DIST=1 timeout 600 python bench.py --steps 50 --warmup 5 --no-cpu-baseline > gpurun_out/${tag}_bench_rccl_world1.json 2> gpurun_out/${tag}_bench_rccl_world1.err
cat gpurun_out/${tag}_pytest_gpu.log | tail -4
tail -2 gpurun_out/${tag}_smoke.log
python - "$tag" <<'PY'
import json, sys
tag = sys.argv[1]
def last(f):
    try: return json.loads(open('gpurun_out/%s_%s.json' % (tag, f)).read().strip().splitlines()[-1])
    except Exception as e: return {'ERR': str(e)}
d = last('bench_config2')
if 'ERR' in d: print('config2', d)
else:
    print('config2 value %.4g ms_per_step %.4f frac %.4f kernel_ms %.4f' % (d['value'], d['ms_per_step'], d['roofline']['frac'], d['roofline']['avg_kernel_ms']))
    print(' e2e', {k: d['end_to_end'][k] for k in ('flatten_s', 'h2d_s', 'reviews_per_s', 'host_threads')})
    print(' cpu', d.get('cpu_baseline', {}).get('value'), d.get('cpu_baseline', {}).get('all_cores'))
    print(' parity', d.get('parity_sample'))
    print(' python', {k: v for k, v in d.get('parity_python_oracle', {}).items() if k != 'checker'})
    print(' totals', d.get('audit_result_totals'))
for f in ('bench_config1', 'bench_config4', 'bench_rccl_world1'):
    d = last(f)
    print(f, 'ERR' in d and d or 'value %.4g ms_per_step %.4f frac %.4f' % (d['value'], d['ms_per_step'], d['roofline']['frac']))
for f in ('stream_closed', 'stream_1M'):
    d = last(f)
    print(f, 'ERR' in d and d or json.dumps(d['stream'])[:700])
try:
    d = json.load(open('gpurun_out/%s_latency.json' % tag))
    for r in d.get('native', {}).get('runs', d.get('runs', [])):
        if r.get('threads') in (1, 64, 256) and r.get('workers') == 2: print('latency', {k: (round(v, 1) if isinstance(v, float) else v) for k, v in r.items() if k in ('threads', 'p50_us', 'p99_us', 'reviews_per_s', 'mean_batch', 'window_us', 'workers', 'max_batch')})
except Exception as e: print('latency ERR', e)
PY
find gpurun_out/${tag}_stats -name '*kernel_stats.csv' | head -1 | xargs -r head -4
cat gpurun_out/${tag}_pmc.log
