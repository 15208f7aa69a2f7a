#!/bin/bash
# round 6, visit t: wave priorities against the age-ordered service of a CU's workgroups (visit s: the k-th workgroup on a CU takes
# 20 / 23 / 28 / 35 k clocks per row group).  GK_PRIO_MODE 1: (k + round) & 3, 2: k, 3: (k - round) & 3
set -u
mkdir -p gpurun_out; export TMPDIR=/tmp
run() {
  timeout 900 python bench.py --config $2 --lean --steps 50 --warmup 5 $3 > gpurun_out/r06t_$1_c$2.json 2> gpurun_out/r06t_$1_c$2.err
  python - gpurun_out/r06t_$1_c$2.json "$1" $2 <<'PY'
import json, sys
try:
    j = json.loads(open(sys.argv[1]).read().strip().split('\n')[-1]); r = j['roofline']
    print('%s config %s: step %.4f ms kernel %.4f ms pairs %s' % (sys.argv[2], sys.argv[3], j['ms_per_step'], r['avg_kernel_ms'], j['config']['global_violating_pairs']))
except Exception as e: print('no line', e)
PY
}
for rep in 1 2; do
  run base$rep 2 ""
  for m in 1 2 3; do GK_JIT_DEFINES="GK_PRIO_MODE=$m" run prio${m}_$rep 2 ""; done
done
run base 4 ""
for m in 1 2 3; do GK_JIT_DEFINES="GK_PRIO_MODE=$m" run prio${m} 4 ""; done
run base_10M 2 "--reviews 10000000"
for m in 1 2 3; do GK_JIT_DEFINES="GK_PRIO_MODE=$m" run prio${m}_10M 2 "--reviews 10000000"; done
GK_JIT_DEFINES="GK_PRIO_MODE=1" GK_KERNEL_PROF=$PWD/gpurun_out/r06t_marks_prio1.bin timeout 600 python bench.py --config 2 --lean --steps 3 --warmup 1 > /dev/null 2> gpurun_out/r06t_prof_prio1.err
GK_JIT_DEFINES="GK_PRIO_MODE=2" GK_KERNEL_PROF=$PWD/gpurun_out/r06t_marks_prio2.bin timeout 600 python bench.py --config 2 --lean --steps 3 --warmup 1 > /dev/null 2> gpurun_out/r06t_prof_prio2.err
