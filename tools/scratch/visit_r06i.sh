#!/bin/bash
# round 6, visit i: geometry / stagger / prefetch re-measured on the leaner kernel (phase 1 is 28 % of a group now, formulas 45 %)
set -u
mkdir -p gpurun_out; export TMPDIR=/tmp
run() {
  timeout 600 python bench.py --config $2 --lean --steps 50 --warmup 5 > gpurun_out/r06i_$1_c$2.json 2> gpurun_out/r06i_$1_c$2.err
  python - gpurun_out/r06i_$1_c$2.json "$1" $2 <<'PY'
import json, sys
try:
    j = json.loads(open(sys.argv[1]).read().strip().split('\n')[-1]); r = j['roofline']
    print('%s config %s: step %.4f ms kernel %.4f ms pairs %s' % (sys.argv[2], sys.argv[3], j['ms_per_step'], r['avg_kernel_ms'], j['config']['global_violating_pairs']))
except Exception as e: print('no line', e)
PY
}
run base 2
GK_STAGGER=0 run stagger0 2
GK_STAGGER=20 run stagger20 2
GK_STAGGER=60 run stagger60 2
GK_JIT_PREFETCH=2 run prefetch2 2
GK_JIT_WAVES=8 GK_JIT_LIST_TRIM=1 run four_groups_per_cu 2
run base 2
run base 1
GK_RPT=128 run rpt128 1
GK_RPT=64 run rpt64 1
GK_RPT=128 GK_JIT_BLOCK=512 run rpt128_block512 1
run base 4
GK_RPT=256 run rpt256 4
GK_RPT=64 run rpt64 4
