#!/bin/bash
# round 6, visit x: wave priorities, fifth pass -- phase bits alone (4: phase 1 high, 20: + outputs / clearing high) against the rotation (7, 23)
# over table sizes up to 10 M objects; GK_PRIO_ROUNDS=1: ranks from one group per workgroup
set -u
mkdir -p gpurun_out; export TMPDIR=/tmp
run() {
  timeout 900 python bench.py --config $2 --lean --steps 50 --warmup 5 $3 > gpurun_out/r06x_$1_c$2.json 2> gpurun_out/r06x_$1_c$2.err
  python - gpurun_out/r06x_$1_c$2.json "$1" $2 <<'PY'
import json, sys
try:
    j = json.loads(open(sys.argv[1]).read().strip().split('\n')[-1]); r = j['roofline']
    print('%s config %s: step %.4f ms kernel %.4f ms pairs %s' % (sys.argv[2], sys.argv[3], j['ms_per_step'], r['avg_kernel_ms'], j['config']['global_violating_pairs']))
except Exception as e: print('no line', e)
PY
}
export GK_PRIO_ROUNDS=1
for n in 300000 600000 1000000 2000000 3000000 5000000 10000000; do
  for m in 0 4 20 7 23; do GK_JIT_PRIO=$m run prio${m}_$n 2 "--reviews $n"; done
done
for rep in 1 2; do
for m in 0 4 20; do GK_JIT_PRIO=$m run prio${m}_$rep 4 ""; done
for m in 0 4 20; do GK_JIT_PRIO=$m run prio${m}_$rep 1 ""; done
done
