#!/bin/bash
# round 6, visit e: EXPERIMENT -- the element marker (P_PRESENT) rides on the rows of the element's `name` member: what the row fusion
# of DESIGN section 11 (i) would buy on configs[2] / [1] / [4], before the flattener learns to guarantee a carrier row per element
set -u
mkdir -p gpurun_out; export TMPDIR=/tmp
run() {
  timeout 600 python bench.py --config $2 --lean --steps 50 --warmup 5 > gpurun_out/r06e_$1_c$2.json 2> gpurun_out/r06e_$1_c$2.err
  python - gpurun_out/r06e_$1_c$2.json "$1" $2 <<'PY'
import json, sys
try:
    j = json.loads(open(sys.argv[1]).read().strip().split('\n')[-1]); r = j['roofline']
    print('%s config %s: step %.4f ms kernel %.4f ms frac %.4f algo %d rows_read %d pairs %s' % (sys.argv[2], sys.argv[3], j['ms_per_step'], r['avg_kernel_ms'], r['frac'], r['algo_bytes_per_launch'], j['config']['rows_read_rank0'], j['config']['global_violating_pairs']))
except Exception as e: print('no line', e)
PY
  grep "gkgpu prof" gpurun_out/r06e_$1_c$2.err | tail -1
}
for cfg in 2 4 1; do
  run base $cfg
  GK_EXPERIMENT_CARRIER=name run carrier $cfg
done
GK_KERNEL_PROF=1 run base_prof 2
GK_KERNEL_PROF=1 GK_EXPERIMENT_CARRIER=name run carrier_prof 2
