#!/bin/bash
# round 6, visit aq: per-constraint totals left by the sweep as ONE ROW PER WORKGROUP (LDS adds in the output stage, plain stores at the
# workgroup's end; the collecting call adds the rows up: no second kernel behind a sweep) against the popcount kernel behind every sweep
# (GK_FUSED_TOTALS=0) -- same box, alternating
set -u
mkdir -p gpurun_out; export TMPDIR=/tmp
run() {
  timeout 900 python bench.py --config $2 --lean --steps ${STEPS:-50} --warmup 5 $3 > gpurun_out/r06aq_$1_c$2.json 2> gpurun_out/r06aq_$1_c$2.err
  python - gpurun_out/r06aq_$1_c$2.json "$1" $2 <<'PY'
import json, sys
try:
    j = json.loads(open(sys.argv[1]).read().strip().split('\n')[-1]); r = j['roofline']
    print('%s config %s: step %.4f ms kernel %.4f ms pairs %s' % (sys.argv[2], sys.argv[3], j['ms_per_step'], r['avg_kernel_ms'], j['config']['global_violating_pairs']))
except Exception as e: print('no line', e)
PY
  grep -v "amdgpu.ids" gpurun_out/r06aq_$1_c$2.err | tail -2
}
for rep in 1 2; do
for cfg in 2 4 1; do
  GK_FUSED_TOTALS=0 run popcount$rep $cfg ""
  run rows$rep $cfg ""
done
done
STEPS=20 GK_FUSED_TOTALS=0 run popcount_20steps 2 ""
STEPS=20 run rows_20steps 2 ""
GK_FUSED_TOTALS=0 run popcount_10M 2 "--reviews 10000000"
run rows_10M 2 "--reviews 10000000"
GK_FUSED_TOTALS=0 run popcount_10M_again 2 "--reviews 10000000"
run rows_10M_again 2 "--reviews 10000000"
bash tools/gpu_visit.sh r06aq benchq
timeout 900 python -m pytest tests/test_parity.py tests/test_result_totals.py tests/test_resident.py tests/test_batcher.py -m gpu -x -q 2>&1 | tail -3
