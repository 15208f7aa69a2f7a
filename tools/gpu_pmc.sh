#!/bin/bash
# PMC counters for the dominant kernel, one counter group per pass (rocprofv3 --pmc with --kernel-trace only).
tag=${1:-pmc}; mkdir -p gpurun_out; export TMPDIR=/tmp
run() { name=$1; shift; rocprofv3 --pmc "$@" --kernel-trace --output-format csv -d gpurun_out/${tag}_$name -o $name -- python bench.py --steps 5 --warmup 1 --no-cpu-baseline $GK_BENCH_ARGS > /dev/null 2> gpurun_out/${tag}_$name.err; }
run sq SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY
run sq2 SQ_INSTS_VMEM_RD SQ_INSTS_SMEM SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_INSTS_VALU SQ_BUSY_CYCLES GRBM_GUI_ACTIVE
run fetch FETCH_SIZE
run write WRITE_SIZE
python - <<'PY'
import csv, glob, collections
for f in sorted(glob.glob('gpurun_out/TAG_*/*counter_collection.csv'.replace('TAG','%s'))):
    pass
PY
for d in sq sq2 fetch write; do
  f=$(find gpurun_out/${tag}_$d -name '*counter_collection.csv' | head -1)
  [ -n "$f" ] && python - "$f" <<'PY'
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
