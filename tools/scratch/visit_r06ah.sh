#!/bin/bash
# round 6, visit ah: the next item's requests issued BEHIND the output stage (the compiler waits vmcnt(0) inside the output stage, i.e. for the rows
# just requested, when they are issued in front of it), and the output stage's LDS reads without the zero defaults (one spilled index less)
set -u
mkdir -p gpurun_out; export TMPDIR=/tmp
run() {
  timeout 900 python bench.py --config $2 --lean --steps 50 --warmup 5 $3 > gpurun_out/r06ah_$1_c$2.json 2> gpurun_out/r06ah_$1_c$2.err
  python - gpurun_out/r06ah_$1_c$2.json "$1" $2 <<'PY'
import json, sys
try:
    j = json.loads(open(sys.argv[1]).read().strip().split('\n')[-1]); r = j['roofline']
    print('%s config %s: step %.4f ms kernel %.4f ms pairs %s' % (sys.argv[2], sys.argv[3], j['ms_per_step'], r['avg_kernel_ms'], j['config']['global_violating_pairs']))
except Exception as e: print('no line', e)
PY
}
S=$PWD/tools/scratch
variants() {
  for rep in 1 2; do
  run base_$1_$rep $2 "$3"
  GK_JIT_BODY_FILE=$S/kernel_body_requests_after_outputs.inc run after_$1_$rep $2 "$3"
  GK_JIT_BODY_FILE=$S/kernel_body_requests_after_outputs_plain_reads.inc run after_plain_$1_$rep $2 "$3"
  GK_JIT_BODY_FILE=$S/kernel_body_plain_reads.inc run plain_$1_$rep $2 "$3"
  done
}
variants 1M 2 ""
variants 3M 2 "--reviews 3000000"
variants 10M 2 "--reviews 10000000"
variants c4 4 ""
variants c1 1 ""
