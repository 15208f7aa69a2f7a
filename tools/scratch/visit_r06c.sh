#!/bin/bash
# round 6, visit c: fused per-constraint totals A/B on one box (configs[2] lean, configs[4], configs[1]); parity of the fused build
set -u
mkdir -p gpurun_out; export TMPDIR=/tmp
for fuse in 1 0 1 0; do
  export GK_JIT_FUSE_COUNTS=$fuse
  for cfg in 2 4 1; do
    timeout 600 python bench.py --config $cfg --lean --steps 50 --warmup 5 > gpurun_out/r06c_c${cfg}_fuse${fuse}.json 2> gpurun_out/r06c_c${cfg}_fuse${fuse}.err
    python - gpurun_out/r06c_c${cfg}_fuse${fuse}.json $cfg $fuse <<'PY'
import json, sys
try:
    j = json.loads(open(sys.argv[1]).read().strip().split('\n')[-1]); r = j['roofline']
    print('config %s fuse %s: step %.4f ms kernel %.4f ms (pair per launch %.4f) frac %.4f pairs %s' % (sys.argv[2], sys.argv[3], j['ms_per_step'], r['avg_kernel_ms'], r['avg_kernel_ms_event_pair_per_launch'], r['frac'], j['config']['global_violating_pairs']))
except Exception as e: print('no line', e)
PY
  done
done
unset GK_JIT_FUSE_COUNTS
bash tools/gpu_visit.sh r06c benchq
