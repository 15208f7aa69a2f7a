#!/bin/bash
# round 6, visit ai: scalar pressure -- the kernel text without the profiling marks, and without the compacted pair list (neither is used by a sweep)
set -u
mkdir -p gpurun_out; export TMPDIR=/tmp
run() {
  timeout 900 python bench.py --config $2 --lean --steps 50 --warmup 5 $3 > gpurun_out/r06ai_$1_c$2.json 2> gpurun_out/r06ai_$1_c$2.err
  python - gpurun_out/r06ai_$1_c$2.json "$1" $2 <<'PY'
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
  GK_JIT_BODY_FILE=$S/kernel_body_no_prof.inc run no_prof_$1_$rep $2 "$3"
  GK_JIT_BODY_FILE=$S/kernel_body_no_prof_no_list.inc run no_prof_no_list_$1_$rep $2 "$3"
  done
}
variants 1M 2 ""
variants 10M 2 "--reviews 10000000"
variants c4 4 ""
variants c1 1 ""
