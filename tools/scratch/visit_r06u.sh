#!/bin/bash
# round 6, visit u: wave priorities, second pass.  GK_PRIO_MODE low two bits: group priority 1 (k + round), 2 k, 3 (k - round) scaled over the
# CU's K workgroups; bit 2: phase 1 at priority 3; bit 3: phase 2 at priority 3
set -u
mkdir -p gpurun_out; export TMPDIR=/tmp
run() {
  timeout 900 python bench.py --config $2 --lean --steps 50 --warmup 5 $3 > gpurun_out/r06u_$1_c$2.json 2> gpurun_out/r06u_$1_c$2.err
  python - gpurun_out/r06u_$1_c$2.json "$1" $2 <<'PY'
import json, sys
try:
    j = json.loads(open(sys.argv[1]).read().strip().split('\n')[-1]); r = j['roofline']
    print('%s config %s: step %.4f ms kernel %.4f ms pairs %s' % (sys.argv[2], sys.argv[3], j['ms_per_step'], r['avg_kernel_ms'], j['config']['global_violating_pairs']))
except Exception as e: print('no line', e)
PY
}
for rep in 1 2; do
  run base$rep 2 ""
  for m in 3 7 11 4 8; do GK_JIT_DEFINES="GK_PRIO_MODE=$m" run prio${m}_$rep 2 ""; done
done
run base 4 ""
for m in 3 7 11 4 8; do GK_JIT_DEFINES="GK_PRIO_MODE=$m" run prio${m} 4 ""; done
run base 1 ""
for m in 3 7 11 4 8; do GK_JIT_DEFINES="GK_PRIO_MODE=$m" run prio${m} 1 ""; done
run base_10M 2 "--reviews 10000000"
for m in 3 7 11; do GK_JIT_DEFINES="GK_PRIO_MODE=$m" run prio${m}_10M 2 "--reviews 10000000"; done
GK_KERNEL_PROF=$PWD/gpurun_out/r06u_marks_c4_base.bin timeout 600 python bench.py --config 4 --lean --steps 3 --warmup 1 > gpurun_out/r06u_prof_c4.json 2> gpurun_out/r06u_prof_c4_base.err
GK_JIT_DEFINES="GK_PRIO_MODE=3" GK_KERNEL_PROF=$PWD/gpurun_out/r06u_marks_c4_prio3.bin timeout 600 python bench.py --config 4 --lean --steps 3 --warmup 1 > /dev/null 2> gpurun_out/r06u_prof_c4_prio3.err
grep -h "gkgpu prof" gpurun_out/r06u_prof_c4_base.err gpurun_out/r06u_prof_c4_prio3.err | tail -2
python - <<'PY'
import json
j = json.loads(open("gpurun_out/r06u_prof_c4.json").read().strip().split('\n')[-1]); print({k: j['roofline'].get(k) for k in ('lds_bytes_per_tile', 'grid', 'block')}, j['config'])
PY
