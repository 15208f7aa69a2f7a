#!/bin/bash
# round 6, visit k: four row groups per CU (64-VGPR budget, trimmed chunk lists) against three, repeated on one box; configs[2] and the 10 M-object table
set -u
mkdir -p gpurun_out; export TMPDIR=/tmp
run() {
  timeout 900 python bench.py --config 2 --lean --steps 50 --warmup 5 $3 > gpurun_out/r06k_$1_$2.json 2> gpurun_out/r06k_$1_$2.err
  python - gpurun_out/r06k_$1_$2.json "$1" $2 <<'PY'
import json, sys
try:
    j = json.loads(open(sys.argv[1]).read().strip().split('\n')[-1]); r = j['roofline']
    print('%s %s: step %.4f ms kernel %.4f ms pairs %s' % (sys.argv[2], sys.argv[3], j['ms_per_step'], r['avg_kernel_ms'], j['config']['global_violating_pairs']))
except Exception as e: print('no line', e)
PY
}
for rep in 1 2 3; do
  run three_per_cu 1M$rep ""
  GK_JIT_WAVES=8 GK_JIT_LIST_TRIM=1 run four_per_cu 1M$rep ""
done
run three_per_cu 10M "--reviews 10000000"
GK_JIT_WAVES=8 GK_JIT_LIST_TRIM=1 run four_per_cu 10M "--reviews 10000000"
