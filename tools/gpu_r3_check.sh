#!/bin/bash
# Round-3 follow-up visit: the driver's own commands (pytest -m gpu WITHOUT GK_JIT_STRICT, smoke) with their exit codes, host
# CPU facts (how many cores the box really grants), the group-tail experiment (983 040 = 5 row groups per workgroup exactly),
# the sharded path on one GPU (RCCL world 1) with enqueued sweeps, the limits probe, RESULT totals on 16 / 64 / 256 host threads
set -u
tag=${1:-r3y}
mkdir -p gpurun_out; export TMPDIR=/tmp
{ echo "nproc $(nproc)"; echo "cpu.max $(cat /sys/fs/cgroup/cpu.max 2>/dev/null)"; lscpu | grep -E 'Model name|Socket|Core|Thread|^CPU\(s\)|NUMA node\(s\)'; } > gpurun_out/${tag}_host.txt 2>&1
timeout 300 python -X faulthandler -c "import __graft_entry__ as g; g.smoke()" > gpurun_out/${tag}_smoke.log 2>&1; echo "smoke rc=$?" >> gpurun_out/${tag}_smoke.log
timeout 1500 python -m pytest tests -m gpu -x -q > gpurun_out/${tag}_pytest_gpu_full.log 2>&1; echo "pytest rc=$?" >> gpurun_out/${tag}_pytest_gpu_full.log
tail -5 gpurun_out/${tag}_pytest_gpu_full.log > gpurun_out/${tag}_pytest_gpu.log
timeout 600 python bench.py --steps 200 --warmup 10 --no-cpu-baseline > gpurun_out/${tag}_bench_1M.json 2> gpurun_out/${tag}_bench_1M.err
timeout 600 python bench.py --steps 200 --warmup 10 --no-cpu-baseline --reviews 983040 > gpurun_out/${tag}_bench_983040.json 2> gpurun_out/${tag}_bench_983040.err
GK_FORCE_DIST=1 timeout 600 python bench.py --steps 50 --warmup 5 --no-cpu-baseline > gpurun_out/${tag}_bench_rccl_world1.json 2> gpurun_out/${tag}_bench_rccl_world1.err; echo "rccl rc=$?" >> gpurun_out/${tag}_bench_rccl_world1.err
timeout 600 python tools/limits_probe.py > gpurun_out/${tag}_limits_probe.json 2> gpurun_out/${tag}_limits_probe.err
timeout 600 python - > gpurun_out/${tag}_totals_threads.json 2> gpurun_out/${tag}_totals_threads.err <<'PY'
import json, os, sys, time
sys.path.insert(0, os.getcwd())
from gatekeeper_amd import driver as D, synth
fx = synth.load_fixtures()
drv = D.Driver(device=0, hostemu=False); c = D.Client(drv)
for t in synth.psp_templates(fx): c.AddTemplate(t)
for k in synth.audit_constraints(): c.AddConstraint(k)
nss = synth.gen_namespaces()
n = 200000
batch = synth.NativeBatch(drv.engine.lib, n, seed=synth.SEED, mixed=True, start=0, namespaces=nss)
table = drv.engine.create_table_native(batch.reviews, n, keep_docs=False, resident=True, keep_text=True)
table.eval()
runs = []
for th in (8, 16, 32, 64, 128, 256):
    os.environ['GK_HOST_THREADS'] = str(th)
    t0 = time.perf_counter(); tot = table.totals(); dt = time.perf_counter() - t0
    runs.append({'threads': th, 'seconds': dt, 'pairs': int(sum(p for _, p in tot.values())), 'results': int(sum(r for r, _ in tot.values()))})
print(json.dumps({'what': 'gk_table_totals (host render of every violating pair) of 200k mixed objects x 50 constraints', 'runs': runs}))
PY
cat gpurun_out/${tag}_host.txt
tail -3 gpurun_out/${tag}_smoke.log
cat gpurun_out/${tag}_pytest_gpu.log
python - "$tag" <<'PY'
import json, sys
tag = sys.argv[1]
def first(f):
    try: return json.loads([l for l in open('gpurun_out/%s_%s.json' % (tag, f)).read().splitlines() if l.startswith('{')][0])
    except Exception as e: return {'ERR': str(e)}
for f in ('bench_1M', 'bench_983040', 'bench_rccl_world1'):
    d = first(f)
    print(f, 'ERR' in d and d or 'value %.4g ms_per_step %.4f frac %.4f kernel_ms %.4f' % (d['value'], d['ms_per_step'], d['roofline']['frac'], d['roofline']['avg_kernel_ms']))
print('limits', json.dumps(first('limits_probe'))[:1500])
print('totals', json.dumps(first('totals_threads'))[:1200])
PY
tail -3 gpurun_out/${tag}_bench_rccl_world1.err
