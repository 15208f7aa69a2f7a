#!/bin/bash
# round 6, visit h: the per-constraint totals beside the next sweep (second stream, two bitmap buffers) A/B; gpu suite on the new default
set -u
mkdir -p gpurun_out; export TMPDIR=/tmp
run() {
  timeout 600 python bench.py --config $2 --lean --steps 50 --warmup 5 > gpurun_out/r06h_$1_c$2.json 2> gpurun_out/r06h_$1_c$2.err
  python - gpurun_out/r06h_$1_c$2.json "$1" $2 <<'PY'
import json, sys
try:
    j = json.loads(open(sys.argv[1]).read().strip().split('\n')[-1]); r = j['roofline']
    print('%s config %s: step %.4f ms kernel %.4f ms pairs %s' % (sys.argv[2], sys.argv[3], j['ms_per_step'], r['avg_kernel_ms'], j['config']['global_violating_pairs']))
except Exception as e: print('no line', e)
PY
}
for rep in 1 2; do
for cfg in 2 4 1; do
  GK_TOTALS_STREAM=0 run same_stream $cfg
  run side_stream $cfg
done
done
bash tools/gpu_visit.sh r06h tests benchq
