#!/bin/bash
# GPU visit f: class dispatch ordered by chunk count (GK_JIT_HOT = classes in the chain; 0 = plain switch), then the records
# of the best setting: parity subset, default bench, rocprofv3 kernel stats, PMC FETCH/WRITE passes
set -u
tag=${1:-run}
mkdir -p gpurun_out; export TMPDIR=/tmp
run() { env "$@" python bench.py --steps 40 --warmup 3 --no-cpu-baseline 2>gpurun_out/${tag}.err | python -c "
import sys, json
d = json.loads(sys.stdin.read().strip().splitlines()[-1])
print('$*  avg_kernel_ms %.4f ms_per_step %.4f frac %.4f pairs %d' % (d['roofline']['avg_kernel_ms'], d['ms_per_step'], d['roofline']['frac'], d['config']['violating_pairs_rank0']))" | tee -a gpurun_out/${tag}_var.log; }
run GK_JIT_HOT=0
run GK_JIT_HOT=4
run GK_JIT_HOT=8
run GK_JIT_HOT=16
run GK_JIT_HOT=0
run GK_JIT_HOT=8
best=$(python - <<PY
import re, collections
t = collections.defaultdict(list)
for l in open('gpurun_out/${tag}_var.log'):
    m = re.match(r'GK_JIT_HOT=(\d+)\s+avg_kernel_ms ([0-9.]+)', l)
    if m: t[int(m.group(1))].append(float(m.group(2)))
print(min(t, key=lambda k: min(t[k])))
PY
)
echo "best GK_JIT_HOT=$best" | tee -a gpurun_out/${tag}_var.log
export GK_JIT_HOT=$best
GK_JIT_STRICT=1 timeout 600 python -m pytest tests -m gpu -x -q -k "row_group or synthetic_parity or edge_cases or config1_demo or corpus" 2>&1 | tail -3 > gpurun_out/${tag}_pytest_gpu.log
timeout 600 python bench.py > gpurun_out/${tag}_bench.json 2> gpurun_out/${tag}_bench.err
(cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/${tag}_stats -o stats -- python $GRAFT_REPO_ROOT/bench.py --steps 50 --warmup 5 --no-cpu-baseline > /dev/null 2> $GRAFT_REPO_ROOT/gpurun_out/${tag}_stats.err)
run_pmc() { name=$1; shift; rocprofv3 --pmc "$@" --kernel-trace --output-format csv -d gpurun_out/${tag}pmc_$name -o $name -- python bench.py --steps 5 --warmup 1 --no-cpu-baseline > /dev/null 2> gpurun_out/${tag}pmc_$name.err; }
run_pmc fetch FETCH_SIZE
run_pmc write WRITE_SIZE
for d in fetch write; do
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
head -c 1300 gpurun_out/${tag}_bench.json; echo
find gpurun_out/${tag}_stats -name '*kernel_stats.csv' | head -1 | xargs -r head -3
cat gpurun_out/${tag}_pmc.log
