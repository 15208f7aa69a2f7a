#!/bin/bash
# round 6, visit ad: 10 M and 3 M objects, alternating builds (every run is its own process and table): no priorities (0), a priority per phase (3003),
# + wave 0's formula share one level up (part0), + GK_JIT_PRE_LIVE=4
set -u
mkdir -p gpurun_out; export TMPDIR=/tmp
run() {
  timeout 900 python bench.py --config $2 --lean --steps 50 --warmup 5 $3 > gpurun_out/r06ad_$1_c$2.json 2> gpurun_out/r06ad_$1_c$2.err
  python - gpurun_out/r06ad_$1_c$2.json "$1" $2 <<'PY'
import json, sys
try:
    j = json.loads(open(sys.argv[1]).read().strip().split('\n')[-1]); r = j['roofline']
    print('%s config %s: step %.4f ms kernel %.4f ms pairs %s' % (sys.argv[2], sys.argv[3], j['ms_per_step'], r['avg_kernel_ms'], j['config']['global_violating_pairs']))
except Exception as e: print('no line', e)
PY
}
for n in 10000000 3000000; do
for rep in 1 2 3; do
  run levels3003_${n}_$rep 2 "--reviews $n"
  GK_JIT_PRIO=0 run none_${n}_$rep 2 "--reviews $n"
  GK_JIT_DEFINES="GK_PRIO_PART0=1" run part0_${n}_$rep 2 "--reviews $n"
  GK_JIT_DEFINES="GK_PRIO_PART0=1" GK_JIT_PRE_LIVE=4 run part0_prelive4_${n}_$rep 2 "--reviews $n"
done
done
