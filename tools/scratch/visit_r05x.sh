# visit r05x (one box): twelve waves per 256-review group (GK_JIT_BLOCK=768: three formula shares per half, two groups per CU = 512 resident
# workgroups -> 3 907 groups are 7.63 -> 8 rounds, 95 % even, against 5.09 -> 6 rounds, 85 %) on configs[2] and configs[1]
set -u; mkdir -p gpurun_out; export TMPDIR=/tmp
run() { tag=$1; shift; timeout 300 python bench.py "$@" --lean --steps 50 --warmup 5 > gpurun_out/r05x_$tag.json 2> gpurun_out/r05x_$tag.err; rc=$?
  python - gpurun_out/r05x_$tag.json $tag $rc <<'PY'
import json, sys
try:
    j = json.loads(open(sys.argv[1]).read().strip().split('\n')[-1]); r = j['roofline']; c = j['config']
    print('%s: step %.4f ms kernel %.4f ms frac %.4f algo %d lds %s pairs %s' % (sys.argv[2], j['ms_per_step'], r['avg_kernel_ms'], r['frac'], r['algo_bytes_per_launch'], r.get('lds_bytes_per_tile'), c.get('global_violating_pairs')))
except Exception as e: print(sys.argv[2], 'rc', sys.argv[3], 'no line', e)
PY
  grep "gkgpu prof" gpurun_out/r05x_$tag.err | tail -1 | cut -c1-260; grep -v "gkgpu prof\|amdgpu.ids" gpurun_out/r05x_$tag.err | tail -1 | cut -c1-200; }
run c2_base
GK_JIT_BLOCK=768 GK_JIT_WAVES=6 GK_PERSIST=2 run c2_b768
GK_JIT_BLOCK=768 GK_JIT_WAVES=6 run c2_b768_nopersist_override
run c2_base2
GK_JIT_BLOCK=768 GK_JIT_WAVES=6 GK_PERSIST=2 run c2_b768_2
GK_KERNEL_PROF=1 GK_JIT_BLOCK=768 GK_JIT_WAVES=6 GK_PERSIST=2 run c2_b768_prof
run c1_base --config 1
GK_JIT_BLOCK=768 GK_JIT_WAVES=6 GK_PERSIST=2 run c1_b768 --config 1
GK_JIT_BLOCK=768 GK_JIT_WAVES=4 GK_PERSIST=2 run c1_b768_w4 --config 1
GK_KERNEL_PROF=1 GK_JIT_BLOCK=768 GK_JIT_WAVES=6 GK_PERSIST=2 run c1_b768_prof --config 1
GK_JIT_BLOCK=1024 GK_JIT_WAVES=4 GK_PERSIST=1 run c1_b1024 --config 1
run c1_base2 --config 1
