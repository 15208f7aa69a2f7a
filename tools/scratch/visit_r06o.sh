#!/bin/bash
# round 6, visit o: the 64-VGPR build spills 14 MB per launch -- fewer preloaded element words kept live (GK_JIT_PRE_LIVE), 72-VGPR budget (7 waves)
set -u
mkdir -p gpurun_out; export TMPDIR=/tmp
run() {
  timeout 900 python bench.py --config 2 --lean --steps 50 --warmup 5 > gpurun_out/r06o_$1.json 2> gpurun_out/r06o_$1.err
  python - gpurun_out/r06o_$1.json "$1" <<'PY'
import json, sys
try:
    j = json.loads(open(sys.argv[1]).read().strip().split('\n')[-1]); r = j['roofline']
    print('%s: step %.4f ms kernel %.4f ms pairs %s' % (sys.argv[2], j['ms_per_step'], r['avg_kernel_ms'], j['config']['global_violating_pairs']))
except Exception as e: print('no line', e)
PY
}
run base1
GK_JIT_PRE_LIVE=8 run pre_live8
GK_JIT_PRE_LIVE=12 run pre_live12
GK_JIT_PRE_LIVE=24 run pre_live24
GK_JIT_PRELOAD=0 run no_preload
GK_JIT_ROLL=0 run no_roll
run base2
GK_JIT_WAVES=7 run waves7
GK_JIT_WAVES=6 run waves6
