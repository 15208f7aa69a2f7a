# visit r05v (one box): four row groups per CU again (GK_JIT_WAVES=8: a 64-VGPR budget; GK_JIT_LIST_TRIM=1: the chunk-list buffers trimmed to
# what the table needs, so that four groups' LDS fits) against three, on configs[2]; GK_PERSIST=4 forces the grid when the LDS estimate says 3
set -u; mkdir -p gpurun_out; export TMPDIR=/tmp
run() { tag=$1; shift; timeout 300 python bench.py "$@" --lean --steps 50 --warmup 5 > gpurun_out/r05v_$tag.json 2> gpurun_out/r05v_$tag.err; rc=$?
  python - gpurun_out/r05v_$tag.json $tag $rc <<'PY'
import json, sys
try:
    j = json.loads(open(sys.argv[1]).read().strip().split('\n')[-1]); r = j['roofline']; c = j['config']
    print('%s: step %.4f ms kernel %.4f ms frac %.4f algo %d lds %s pairs %s' % (sys.argv[2], j['ms_per_step'], r['avg_kernel_ms'], r['frac'], r['algo_bytes_per_launch'], r.get('lds_bytes_per_tile'), c.get('global_violating_pairs')))
except Exception as e: print(sys.argv[2], 'rc', sys.argv[3], 'no line', e)
PY
  grep "gkgpu prof" gpurun_out/r05v_$tag.err | tail -1 | cut -c1-260; grep -v "gkgpu prof\|amdgpu.ids" gpurun_out/r05v_$tag.err | tail -1 | cut -c1-200; }
run c2_base
GK_JIT_WAVES=8 GK_JIT_LIST_TRIM=1 run c2_w8trim
GK_JIT_WAVES=8 GK_JIT_LIST_TRIM=1 GK_PERSIST=4 run c2_w8trim_p4
GK_JIT_LIST_TRIM=1 run c2_trim_only
GK_JIT_WAVES=7 run c2_w7
run c2_base2
GK_JIT_WAVES=8 GK_JIT_LIST_TRIM=1 run c2_w8trim2
GK_KERNEL_PROF=1 GK_JIT_WAVES=8 GK_JIT_LIST_TRIM=1 run c2_w8trim_prof
GK_KERNEL_PROF=1 run c2_base_prof
run c1_base --config 1
GK_JIT_WAVES=8 GK_JIT_LIST_TRIM=1 run c1_w8trim --config 1
GK_PERSIST=2 run c1_p2 --config 1
