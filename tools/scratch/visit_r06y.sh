#!/bin/bash
# round 6, visit y: a wave priority per PHASE (GK_PRIO_LEVELS = phase 1 | bounds | formulas | outputs + requests + clearing), variants
set -u
mkdir -p gpurun_out; export TMPDIR=/tmp
run() {
  timeout 900 python bench.py --config $2 --lean --steps 50 --warmup 5 $3 > gpurun_out/r06y_$1_c$2.json 2> gpurun_out/r06y_$1_c$2.err
  python - gpurun_out/r06y_$1_c$2.json "$1" $2 <<'PY'
import json, sys
try:
    j = json.loads(open(sys.argv[1]).read().strip().split('\n')[-1]); r = j['roofline']
    print('%s config %s: step %.4f ms kernel %.4f ms pairs %s' % (sys.argv[2], sys.argv[3], j['ms_per_step'], r['avg_kernel_ms'], j['config']['global_violating_pairs']))
except Exception as e: print('no line', e)
PY
}
for n in 1000000 3000000 10000000; do
  for m in 0 3003 3303 2003 3002 3001 1003 330; do GK_JIT_PRIO=$m run levels${m}_$n 2 "--reviews $n"; done
done
for m in 0 3003 3303; do GK_JIT_PRIO=$m run levels${m}_again_1000000 2 ""; done
for m in 0 3003 3303 2003; do GK_JIT_PRIO=$m run levels${m} 4 ""; done
for m in 0 3003 3303 2003; do GK_JIT_PRIO=$m run levels${m} 1 ""; done
