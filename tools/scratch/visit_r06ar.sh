#!/bin/bash
# round 6, visit ar: variants of the per-workgroup totals rows in the sweep (where the LDS add sits, where the row's address waits)
# through GK_JIT_BODY_FILE, against the popcount kernel behind every sweep (GK_FUSED_TOTALS=0) -- same box, alternating
set -u
mkdir -p gpurun_out; export TMPDIR=/tmp
run() {
  timeout 900 python bench.py --config $2 --lean --steps ${STEPS:-50} --warmup 5 $3 > gpurun_out/r06ar_$1_c$2.json 2> gpurun_out/r06ar_$1_c$2.err
  python - gpurun_out/r06ar_$1_c$2.json "$1" $2 <<'PY'
import json, sys
try:
    j = json.loads(open(sys.argv[1]).read().strip().split('\n')[-1]); r = j['roofline']
    print('%s config %s: step %.4f ms kernel %.4f ms pairs %s' % (sys.argv[2], sys.argv[3], j['ms_per_step'], r['avg_kernel_ms'], j['config']['global_violating_pairs']))
except Exception as e: print('no line', e)
PY
  grep -v "amdgpu.ids" gpurun_out/r06ar_$1_c$2.err | tail -2
}
for rep in 1 2; do
for cfg in 2 1 4; do
  GK_FUSED_TOTALS=0 run popcount$rep $cfg ""
  run A$rep $cfg ""
  for v in B C D E; do GK_JIT_BODY_FILE=$PWD/tools/scratch/kernel_body_totals_$v.inc run $v$rep $cfg ""; done
done
done
for rep in 1 2; do
  GK_FUSED_TOTALS=0 run popcount_10M_$rep 2 "--reviews 10000000"
  for v in B D; do GK_JIT_BODY_FILE=$PWD/tools/scratch/kernel_body_totals_$v.inc run ${v}_10M_$rep 2 "--reviews 10000000"; done
done
