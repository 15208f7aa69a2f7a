#!/bin/bash
# round 6, visit v: wave priorities, third pass -- from how many row groups per workgroup does the rotation pay (GK_PRIO_ROUNDS), and
# mode bit 4 (outputs + clearing at the top priority)
set -u
mkdir -p gpurun_out; export TMPDIR=/tmp
run() {
  timeout 900 python bench.py --config $2 --lean --steps 50 --warmup 5 $3 > gpurun_out/r06v_$1_c$2.json 2> gpurun_out/r06v_$1_c$2.err
  python - gpurun_out/r06v_$1_c$2.json "$1" $2 <<'PY'
import json, sys
try:
    j = json.loads(open(sys.argv[1]).read().strip().split('\n')[-1]); r = j['roofline']
    print('%s config %s: step %.4f ms kernel %.4f ms pairs %s' % (sys.argv[2], sys.argv[3], j['ms_per_step'], r['avg_kernel_ms'], j['config']['global_violating_pairs']))
except Exception as e: print('no line', e)
PY
}
export GK_PRIO_ROUNDS=1
for n in 300000 400000 500000 600000 800000; do
  run base_$n 2 "--reviews $n"
  GK_JIT_DEFINES="GK_PRIO_MODE=7" run prio7_$n 2 "--reviews $n"
  GK_JIT_DEFINES="GK_PRIO_MODE=4" run prio4_$n 2 "--reviews $n"
done
for rep in 1 2; do
  run base$rep 2 ""
  for m in 7 23 20 3; do GK_JIT_DEFINES="GK_PRIO_MODE=$m" run prio${m}_$rep 2 ""; done
done
run base 4 ""
for m in 4 20; do GK_JIT_DEFINES="GK_PRIO_MODE=$m" run prio${m} 4 ""; done
GK_PRIO_ROUNDS=3 GK_JIT_DEFINES="GK_PRIO_MODE=7" run prio7_gated 4 ""
run base 1 ""
for m in 4 20; do GK_JIT_DEFINES="GK_PRIO_MODE=$m" run prio${m} 1 ""; done
GK_PRIO_ROUNDS=3 GK_JIT_DEFINES="GK_PRIO_MODE=7" run prio7_gated 1 ""
