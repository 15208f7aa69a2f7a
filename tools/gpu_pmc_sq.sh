#!/bin/bash
# SQ instruction / wait counters of the dominant kernel for the current environment (one rocprofv3 --pmc pass)
tag=$1; mkdir -p gpurun_out; export TMPDIR=/tmp
rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY --kernel-trace --output-format csv -d gpurun_out/${tag} -o sq -- python bench.py --steps 5 --warmup 1 --no-cpu-baseline > /dev/null 2> gpurun_out/${tag}.err
f=$(find gpurun_out/${tag} -name '*counter_collection.csv' | head -1)
python - "$f" "$tag" <<'PY'
import csv, sys, collections
acc = collections.defaultdict(float); n = collections.Counter()
for r in csv.DictReader(open(sys.argv[1])):
    if 'tiles' not in r.get('Kernel_Name', ''): continue
    acc[r['Counter_Name']] += float(r['Counter_Value']); n[r['Counter_Name']] += 1
print(sys.argv[2], ' '.join('%s=%.0f' % (c, v / n[c]) for c, v in sorted(acc.items())))
PY
