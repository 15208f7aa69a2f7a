#!/bin/bash
# round 6, visit d: what makes the fused per-constraint totals slow -- the fences or the contended atomics
set -u
mkdir -p gpurun_out; export TMPDIR=/tmp
run() {
  timeout 600 python bench.py --config 2 --lean --steps 50 --warmup 5 > gpurun_out/r06d_$1.json 2> gpurun_out/r06d_$1.err
  python - gpurun_out/r06d_$1.json "$1" <<'PY'
import json, sys
try:
    j = json.loads(open(sys.argv[1]).read().strip().split('\n')[-1]); r = j['roofline']
    print('%s: step %.4f ms kernel %.4f ms frac %.4f pairs %s' % (sys.argv[2], j['ms_per_step'], r['avg_kernel_ms'], r['frac'], j['config']['global_violating_pairs']))
except Exception as e: print('no line', e)
PY
}
GK_JIT_FUSE_COUNTS=0 run unfused
GK_JIT_DEFINES="GK_CNT_NOFENCE" run fused_nofence
GK_JIT_DEFINES="GK_CNT_PER_XCD" run fused_per_xcd
GK_JIT_DEFINES="GK_CNT_PER_XCD;GK_CNT_NOFENCE" run fused_per_xcd_nofence
run fused
