#!/bin/bash
# round 6, visit am: geometry and generator knobs re-measured on the final kernel (priorities per phase, no profiling aids)
set -u
mkdir -p gpurun_out; export TMPDIR=/tmp
run() {
  timeout 900 python bench.py --config $2 --lean --steps 50 --warmup 5 $3 > gpurun_out/r06am_$1_c$2.json 2> gpurun_out/r06am_$1_c$2.err
  python - gpurun_out/r06am_$1_c$2.json "$1" $2 <<'PY'
import json, sys
try:
    j = json.loads(open(sys.argv[1]).read().strip().split('\n')[-1]); r = j['roofline']
    print('%s config %s: step %.4f ms kernel %.4f ms pairs %s' % (sys.argv[2], sys.argv[3], j['ms_per_step'], r['avg_kernel_ms'], j['config']['global_violating_pairs']))
except Exception as e: print('no line', e)
PY
}
for n in 1000000 10000000; do
  run base_$n 2 "--reviews $n"
  GK_RPT=128 run rpt128_$n 2 "--reviews $n"
  GK_RPT=512 run rpt512_$n 2 "--reviews $n"
  GK_JIT_ROLL=0 run no_roll_$n 2 "--reviews $n"
  GK_JIT_CONJ=0 run no_conj_$n 2 "--reviews $n"
  GK_JIT_RES_LANES=0 run no_res_lanes_$n 2 "--reviews $n"
  GK_JIT_LIST_TRIM=0 run no_list_trim_$n 2 "--reviews $n"
  GK_JIT_WAVES=7 run waves7_$n 2 "--reviews $n"
  run base_again_$n 2 "--reviews $n"
done
run base 1 ""
GK_RPT=128 run rpt128 1 ""
GK_RPT=64 run rpt64 1 ""
GK_RPT=128 GK_JIT_BLOCK=512 run rpt128_block512 1 ""
