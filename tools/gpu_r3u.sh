#!/bin/bash
# Round-3 visit u: final records of round 3 (count kernels without clearing passes, split masks, deep dictionary expressions): gpu test suite, smoke,
# the default bench line (configs[2]), rocprofv3 kernel stats and the PMC passes (SQ, FETCH_SIZE, WRITE_SIZE: separate passes)
set -u
tag=${1:-r3u}
mkdir -p gpurun_out; export TMPDIR=/tmp
GK_JIT_STRICT=1 timeout 1200 python -m pytest tests -m gpu -x -q 2>&1 | tail -6 > gpurun_out/${tag}_pytest_gpu.log
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" > gpurun_out/${tag}_smoke.log 2>&1
timeout 900 python bench.py > gpurun_out/${tag}_bench_config2.json 2> gpurun_out/${tag}_bench_config2.err
(cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/${tag}_stats -o stats -- python $GRAFT_REPO_ROOT/bench.py --steps 50 --warmup 5 --no-cpu-baseline --lean > /dev/null 2> $GRAFT_REPO_ROOT/gpurun_out/${tag}_stats.err)
run_pmc() { name=$1; shift; timeout 400 rocprofv3 --pmc "$@" --kernel-trace --output-format csv -d gpurun_out/${tag}pmc_$name -o $name -- python bench.py --steps 5 --warmup 1 --no-cpu-baseline --lean > /dev/null 2> gpurun_out/${tag}pmc_$name.err; }
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
tail -3 gpurun_out/${tag}_pytest_gpu.log
tail -2 gpurun_out/${tag}_smoke.log
tail -1 gpurun_out/${tag}_bench_config2.json | cut -c1-900
find gpurun_out/${tag}_stats -name '*kernel_stats.csv' | head -1 | xargs -r head -5
cat gpurun_out/${tag}_pmc.log
