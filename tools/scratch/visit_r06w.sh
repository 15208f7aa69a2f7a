#!/bin/bash
# round 6, visit w: wave priorities, fourth pass -- the packed-state build; rotation (7) against rotation + longest-remaining-first (39)
# against none (GK_JIT_PRIO=0) and phase 1 alone (4), over table sizes (row groups per workgroup 1.1 .. 38); GK_PRIO_ROUNDS=1: ranks from one group per workgroup
set -u
mkdir -p gpurun_out; export TMPDIR=/tmp
run() {
  timeout 900 python bench.py --config $2 --lean --steps 50 --warmup 5 $3 > gpurun_out/r06w_$1_c$2.json 2> gpurun_out/r06w_$1_c$2.err
  python - gpurun_out/r06w_$1_c$2.json "$1" $2 <<'PY'
import json, sys
try:
    j = json.loads(open(sys.argv[1]).read().strip().split('\n')[-1]); r = j['roofline']
    print('%s config %s: step %.4f ms kernel %.4f ms pairs %s' % (sys.argv[2], sys.argv[3], j['ms_per_step'], r['avg_kernel_ms'], j['config']['global_violating_pairs']))
except Exception as e: print('no line', e)
PY
}
export GK_PRIO_ROUNDS=1
for n in 300000 400000 500000 600000 700000 800000 900000 1000000 1200000 1500000 2000000; do
  for m in 0 4 7 39; do GK_JIT_PRIO=$m run prio${m}_$n 2 "--reviews $n"; done
done
for m in 0 7 39; do GK_JIT_PRIO=$m run prio${m}_10M 2 "--reviews 10000000"; done
for m in 0 4 7 39; do GK_JIT_PRIO=$m run prio${m} 4 ""; done
for m in 0 4 7 39; do GK_JIT_PRIO=$m run prio${m} 1 ""; done
for m in 0 39; do GK_JIT_PRIO=$m run prio${m}_again_1000000 2 ""; done
