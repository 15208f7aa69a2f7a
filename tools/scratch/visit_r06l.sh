#!/bin/bash
# round 6, visit l: trimmed chunk lists (a fourth row group per CU where it fits) as the default -- configs[1] and the corpus checked, gpu suite
set -u
mkdir -p gpurun_out; export TMPDIR=/tmp
run() {
  timeout 900 python bench.py --config $2 --lean --steps 50 --warmup 5 > gpurun_out/r06l_$1_c$2.json 2> gpurun_out/r06l_$1_c$2.err
  python - gpurun_out/r06l_$1_c$2.json "$1" $2 <<'PY'
import json, sys
try:
    j = json.loads(open(sys.argv[1]).read().strip().split('\n')[-1]); r = j['roofline']
    print('%s config %s: step %.4f ms kernel %.4f ms pairs %s' % (sys.argv[2], sys.argv[3], j['ms_per_step'], r['avg_kernel_ms'], j['config']['global_violating_pairs']))
except Exception as e: print('no line', e)
PY
}
for rep in 1 2; do
for cfg in 1 4 2; do
  GK_JIT_LIST_TRIM=0 run full_lists$rep $cfg
  run trimmed$rep $cfg
done
done
bash tools/gpu_visit.sh r06l tests benchq
