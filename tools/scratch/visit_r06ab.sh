#!/bin/bash
# round 6, visit ab: on top of the priority per phase -- the formula share that writes the violation words one level up, stagger, two chunks in flight,
# three row groups per CU at 80 VGPRs
set -u
mkdir -p gpurun_out; export TMPDIR=/tmp
run() {
  timeout 900 python bench.py --config $2 --lean --steps 50 --warmup 5 $3 > gpurun_out/r06ab_$1_c$2.json 2> gpurun_out/r06ab_$1_c$2.err
  python - gpurun_out/r06ab_$1_c$2.json "$1" $2 <<'PY'
import json, sys
try:
    j = json.loads(open(sys.argv[1]).read().strip().split('\n')[-1]); r = j['roofline']
    print('%s config %s: step %.4f ms kernel %.4f ms pairs %s' % (sys.argv[2], sys.argv[3], j['ms_per_step'], r['avg_kernel_ms'], j['config']['global_violating_pairs']))
except Exception as e: print('no line', e)
PY
}
for n in 1000000 3000000; do
  run base_$n 2 "--reviews $n"
  GK_JIT_DEFINES="GK_PRIO_PART0=1" run part0_up_$n 2 "--reviews $n"
  GK_JIT_PRIO=3013 GK_JIT_DEFINES="GK_PRIO_PART0=0" run part1_up_$n 2 "--reviews $n"
  GK_STAGGER=20 run stagger20_$n 2 "--reviews $n"
  GK_STAGGER=40 run stagger40_$n 2 "--reviews $n"
  GK_JIT_PREFETCH=2 run prefetch2_$n 2 "--reviews $n"
  GK_PERSIST=3 GK_JIT_WAVES=6 run three_per_cu_$n 2 "--reviews $n"
  GK_JIT_PRE_LIVE=8 run pre_live8_$n 2 "--reviews $n"
  run base_again_$n 2 "--reviews $n"
done
