#!/bin/bash
# round 6, visit ae: preloaded element words kept live per formula part (GK_JIT_PRE_LIVE, default 16) on the build with priorities
set -u
mkdir -p gpurun_out; export TMPDIR=/tmp
run() {
  timeout 900 python bench.py --config $2 --lean --steps 50 --warmup 5 $3 > gpurun_out/r06ae_$1_c$2.json 2> gpurun_out/r06ae_$1_c$2.err
  python - gpurun_out/r06ae_$1_c$2.json "$1" $2 <<'PY'
import json, sys
try:
    j = json.loads(open(sys.argv[1]).read().strip().split('\n')[-1]); r = j['roofline']
    print('%s config %s: step %.4f ms kernel %.4f ms pairs %s' % (sys.argv[2], sys.argv[3], j['ms_per_step'], r['avg_kernel_ms'], j['config']['global_violating_pairs']))
except Exception as e: print('no line', e)
PY
}
sweep() {
  for rep in 1 2; do
  for pl in 16 8 6 4 2 1; do GK_JIT_PRE_LIVE=$pl run prelive${pl}_$1_$rep $2 "$3"; done
  GK_JIT_PRELOAD=0 run no_preload_$1_$rep $2 "$3"
  done
}
sweep 1M 2 ""
sweep 10M 2 "--reviews 10000000"
sweep c4 4 ""
sweep c1 1 ""
