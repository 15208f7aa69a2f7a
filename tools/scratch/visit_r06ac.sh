#!/bin/bash
# round 6, visit ac: the formula share of wave 0 of each half (it also writes the violation words) one priority level above the other share(s);
# with and without fewer preloaded element words kept live (GK_JIT_PRE_LIVE=8)
set -u
mkdir -p gpurun_out; export TMPDIR=/tmp
run() {
  timeout 900 python bench.py --config $2 --lean --steps 50 --warmup 5 $3 > gpurun_out/r06ac_$1_c$2.json 2> gpurun_out/r06ac_$1_c$2.err
  python - gpurun_out/r06ac_$1_c$2.json "$1" $2 <<'PY'
import json, sys
try:
    j = json.loads(open(sys.argv[1]).read().strip().split('\n')[-1]); r = j['roofline']
    print('%s config %s: step %.4f ms kernel %.4f ms pairs %s' % (sys.argv[2], sys.argv[3], j['ms_per_step'], r['avg_kernel_ms'], j['config']['global_violating_pairs']))
except Exception as e: print('no line', e)
PY
}
variants() {
  run base_$1 $2 "$3"
  GK_JIT_DEFINES="GK_PRIO_PART0=1" run part0_$1 $2 "$3"
  GK_JIT_DEFINES="GK_PRIO_PART0=1" GK_JIT_PRE_LIVE=8 run part0_prelive8_$1 $2 "$3"
  GK_JIT_DEFINES="GK_PRIO_PART0=1" GK_JIT_PRE_LIVE=4 run part0_prelive4_$1 $2 "$3"
  GK_JIT_PRIO=3023 GK_JIT_DEFINES="GK_PRIO_PART0=1" run part0_is_low_$1 $2 "$3"
}
variants 1M 2 ""
variants 10M 2 "--reviews 10000000"
variants 600k 2 "--reviews 600000"
variants c4 4 ""
variants c1 1 ""
variants 1M_again 2 ""
