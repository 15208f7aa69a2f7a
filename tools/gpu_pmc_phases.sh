#!/bin/bash
# instruction counts of the dominant kernel per phase-isolation setting (GK_DBG_PHASE 0..3), one rocprofv3 --pmc pass each
tag=${1:-pmcph}; mkdir -p gpurun_out; export TMPDIR=/tmp
for ph in 0 1 2 3; do
  GK_DBG_PHASE=$ph rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_ACTIVE_INST_ANY --kernel-trace --output-format csv -d gpurun_out/${tag}_$ph -o p$ph -- python bench.py --steps 5 --warmup 1 --no-cpu-baseline > /dev/null 2> gpurun_out/${tag}_$ph.err
  f=$(find gpurun_out/${tag}_$ph -name '*counter_collection.csv' | head -1)
  [ -n "$f" ] && python - "$f" $ph <<'PY' | tee -a gpurun_out/${tag}_summary.log
import csv, sys, collections
acc = collections.defaultdict(float); n = collections.Counter()
for r in csv.DictReader(open(sys.argv[1])):
    if 'tiles' not in r.get('Kernel_Name', ''): continue
    acc[r['Counter_Name']] += float(r['Counter_Value']); n[r['Counter_Name']] += 1
print('GK_DBG_PHASE=%s ' % sys.argv[2] + ' '.join('%s=%.0f' % (c, v / n[c]) for c, v in sorted(acc.items())))
PY
done
