#!/bin/bash
# round 6, visit au: the totals rows with the row's address parked in LDS as an INTEGER and cast to a global pointer at the workgroup's
# end (variant F, through GK_JIT_BODY_FILE) against the default text (address from the kernel arguments and blockIdx.x at the end) and
# the popcount kernel behind every sweep (GK_FUSED_TOTALS=0) -- same box, alternating, 1 M and 10 M objects
set -u
mkdir -p gpurun_out; export TMPDIR=/tmp
run() {
  timeout 900 python bench.py --config $2 --lean --steps 50 --warmup 5 $3 > gpurun_out/r06au_$1_c$2.json 2> gpurun_out/r06au_$1_c$2.err
  python - gpurun_out/r06au_$1_c$2.json "$1" $2 <<'PY'
import json, sys
try:
    j = json.loads(open(sys.argv[1]).read().strip().split('\n')[-1]); r = j['roofline']
    print('%s config %s: step %.4f ms kernel %.4f ms pairs %s' % (sys.argv[2], sys.argv[3], j['ms_per_step'], r['avg_kernel_ms'], j['config']['global_violating_pairs']))
except Exception as e: print('no line', e)
PY
  grep -v "amdgpu.ids" gpurun_out/r06au_$1_c$2.err | tail -2
}
F=$PWD/tools/scratch/kernel_body_totals_F.inc
for rep in 1 2; do
for cfg in 2 1 4; do
  GK_FUSED_TOTALS=0 run popcount$rep $cfg ""
  run default$rep $cfg ""
  GK_JIT_BODY_FILE=$F run F$rep $cfg ""
done
done
for rep in 1 2 3; do
  GK_FUSED_TOTALS=0 run popcount_10M_$rep 2 "--reviews 10000000"
  run default_10M_$rep 2 "--reviews 10000000"
  GK_JIT_BODY_FILE=$F run F_10M_$rep 2 "--reviews 10000000"
done
