#!/bin/bash
# round 6, visit q: 128-review row groups (8 per CU, 4 waves each) for configs[2] / [3] against the default 256-review groups (4 per CU, 8 waves each)
set -u
mkdir -p gpurun_out; export TMPDIR=/tmp
run() {
  timeout 900 python bench.py --config 2 --lean --steps 50 --warmup 5 $2 > gpurun_out/r06q_$1.json 2> gpurun_out/r06q_$1.err
  python - gpurun_out/r06q_$1.json "$1" <<'PY'
import json, sys
try:
    j = json.loads(open(sys.argv[1]).read().strip().split('\n')[-1]); r = j['roofline']
    print('%s: step %.4f ms kernel %.4f ms lds %s pairs %s' % (sys.argv[2], j['ms_per_step'], r['avg_kernel_ms'], r['lds_bytes_per_tile'], j['config']['global_violating_pairs']))
except Exception as e: print('no line', e)
PY
}
run rpt256_1M ""
GK_RPT=128 run rpt128_1M ""
GK_RPT=128 GK_JIT_BLOCK=512 run rpt128_block512_1M ""
GK_RPT=512 run rpt512_1M ""
run rpt256_1M_again ""
GK_RPT=128 run rpt128_1M_again ""
run rpt256_10M "--reviews 10000000"
GK_RPT=128 run rpt128_10M "--reviews 10000000"
