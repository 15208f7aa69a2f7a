# visit r05z (one box): the chunk loop with two slots that swap roles (tools/scratch/kernel_body_pingpong.inc through GK_JIT_BODY_FILE,
# GK_JIT_DEFINES=GK_PINGPONG=1: no slot copies at the back edge, an explicit wait in front of the next request) against the product body
set -u; mkdir -p gpurun_out; export TMPDIR=/tmp
PP=$PWD/tools/scratch/kernel_body_pingpong.inc   # (kept as a patch: patch -o tools/scratch/kernel_body_pingpong.inc gatekeeper_amd/csrc/kernel_body.inc tools/scratch/kernel_body_pingpong.patch)
run() { tag=$1; shift; timeout 300 python bench.py "$@" --lean --steps 50 --warmup 5 > gpurun_out/r05z_$tag.json 2> gpurun_out/r05z_$tag.err; rc=$?
  python - gpurun_out/r05z_$tag.json $tag $rc <<'PY'
import json, sys
try:
    j = json.loads(open(sys.argv[1]).read().strip().split('\n')[-1]); r = j['roofline']; c = j['config']
    print('%s: step %.4f ms kernel %.4f ms frac %.4f algo %d pairs %s' % (sys.argv[2], j['ms_per_step'], r['avg_kernel_ms'], r['frac'], r['algo_bytes_per_launch'], c.get('global_violating_pairs')))
except Exception as e: print(sys.argv[2], 'rc', sys.argv[3], 'no line', e)
PY
  grep "gkgpu prof" gpurun_out/r05z_$tag.err | tail -1 | cut -c1-260; grep -v "gkgpu prof\|amdgpu.ids" gpurun_out/r05z_$tag.err | tail -1 | cut -c1-200; }
run c2_base
GK_JIT_BODY_FILE=$PP GK_JIT_DEFINES="GK_PINGPONG=1" run c2_pp
run c2_base2
GK_JIT_BODY_FILE=$PP GK_JIT_DEFINES="GK_PINGPONG=1" run c2_pp2
GK_KERNEL_PROF=1 GK_JIT_BODY_FILE=$PP GK_JIT_DEFINES="GK_PINGPONG=1" run c2_pp_prof
run c1_base --config 1
GK_JIT_BODY_FILE=$PP GK_JIT_DEFINES="GK_PINGPONG=1" run c1_pp --config 1
run c4_base --config 4
GK_JIT_BODY_FILE=$PP GK_JIT_DEFINES="GK_PINGPONG=1" run c4_pp --config 4
run c4_base2 --config 4
GK_JIT_BODY_FILE=$PP GK_JIT_DEFINES="GK_PINGPONG=1" run c4_pp2 --config 4
