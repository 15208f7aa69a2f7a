#!/bin/bash
# Round-3 experiments: issue priority of the streaming / formula waves (GK_PRIO_K through GK_JIT_DEFINES), host pool size under
# the box's CPU quota (ingest of the 1 M-object table on 16 / 32 / 64 threads), kernel trace of the sharded sweep (what the
# exchange step costs on the device against the host), streaming with the quota-sized pool
set -u
tag=${1:-r3x}
mkdir -p gpurun_out; export TMPDIR=/tmp
for v in 0 1 2; do
  if [ $v = 0 ]; then defs=""; else defs="GK_PRIO_K=$v"; fi
  GK_JIT_DEFINES="$defs" timeout 600 python bench.py --steps 200 --warmup 10 --no-cpu-baseline > gpurun_out/${tag}_bench_prio$v.json 2> gpurun_out/${tag}_bench_prio$v.err
done
timeout 900 python - > gpurun_out/${tag}_ingest_threads.json 2> gpurun_out/${tag}_ingest_threads.err <<'PY'
import json, os, sys, time
sys.path.insert(0, os.getcwd())
from gatekeeper_amd import driver as D, synth
fx = synth.load_fixtures()
drv = D.Driver(device=0, hostemu=False); c = D.Client(drv)
for t in synth.psp_templates(fx): c.AddTemplate(t)
for k in synth.audit_constraints(): c.AddConstraint(k)
nss = synth.gen_namespaces()
n = 1000000
batch = synth.NativeBatch(drv.engine.lib, n, seed=synth.SEED, mixed=True, start=0, namespaces=nss)
runs = []
for th in (64, 16, 24, 32, 64, 16, 128):
    os.environ['GK_HOST_THREADS'] = str(th)
    t0 = time.perf_counter()
    table = drv.engine.create_table_native(batch.reviews, n, keep_docs=False, resident=True)
    dt = time.perf_counter() - t0
    st = table.stats()
    runs.append({'threads': th, 'create_s': dt, 'flatten_s': st['flatten_s'], 'h2d_s': st['upload_s'], 'reviews_per_s': n / (st['flatten_s'] + st['upload_s'])})
    table.free()
print(json.dumps({'what': 'gk_table_create of 1 M mixed objects (634 MB of JSON text) by host pool size; first run warms the staging pools', 'host_cpus': int(drv.engine.lib.gk_host_cpus()), 'runs': runs}))
PY
(cd /tmp && GK_FORCE_DIST=1 timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/${tag}_dist_stats -o stats -- python $GRAFT_REPO_ROOT/bench.py --steps 50 --warmup 5 --no-cpu-baseline > $GRAFT_REPO_ROOT/gpurun_out/${tag}_bench_rccl_world1_prof.json 2> $GRAFT_REPO_ROOT/gpurun_out/${tag}_dist_stats.err)
timeout 600 python bench.py --config 4 --streaming --offered 0 --stream-batches 16 > gpurun_out/${tag}_stream_closed.json 2> gpurun_out/${tag}_stream_closed.err
timeout 600 python tools/latency_probe.py > gpurun_out/${tag}_latency.json 2> gpurun_out/${tag}_latency.err
python - "$tag" <<'PY'
import json, sys
tag = sys.argv[1]
def first(f):
    try: return json.loads([l for l in open('gpurun_out/%s_%s.json' % (tag, f)).read().splitlines() if l.startswith('{')][0])
    except Exception as e: return {'ERR': str(e)}
for f in ('bench_prio0', 'bench_prio1', 'bench_prio2', 'bench_rccl_world1_prof'):
    d = first(f)
    print(f, 'ERR' in d and d or 'value %.4g ms_per_step %.4f frac %.4f kernel_ms %.4f e2e flatten %.3f threads %s totals %s' % (d['value'], d['ms_per_step'], d['roofline']['frac'], d['roofline']['avg_kernel_ms'], d['end_to_end']['flatten_s'], d['end_to_end']['host_threads'], d.get('audit_result_totals', {}).get('seconds')))
print('ingest', json.dumps(first('ingest_threads'))[:1500])
d = first('stream_closed'); print('stream', 'ERR' in d and d or json.dumps(d['stream'])[:600])
try:
    d = json.load(open('gpurun_out/%s_latency.json' % tag))
    for r in d.get('native', {}).get('runs', []):
        if r.get('threads') in (1, 64, 256) and r.get('workers') == 2: print('latency', {k: (round(v, 1) if isinstance(v, float) else v) for k, v in r.items() if k in ('threads', 'p50_us', 'p99_us', 'reviews_per_s', 'mean_batch', 'window_us', 'workers', 'max_batch')})
except Exception as e: print('latency ERR', e)
PY
find gpurun_out/${tag}_dist_stats -name '*kernel_stats.csv' | head -1 | xargs -r head -12
