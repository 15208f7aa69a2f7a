# the 200-template corpus with smaller plan groups (GK_GROUP_MAX): smaller code objects / fewer accumulator words against more walks of the table
set -u; mkdir -p gpurun_out
run() { tag=$1; timeout 300 python bench.py --config 4 --lean --steps 50 --warmup 5 > gpurun_out/r4v_$tag.json 2> gpurun_out/r4v_$tag.err; rc=$?
  python - gpurun_out/r4v_$tag.json $tag $rc <<'PY'
import json, sys
try:
    j = json.loads(open(sys.argv[1]).read().strip().split('\n')[-1]); r = j['roofline']; c = j['config']
    print(sys.argv[2], 'rc', sys.argv[3], 'step %.4f ms sum-of-group-kernels %.4f ms lds %s rows_read %s' % (j['ms_per_step'], r['avg_kernel_ms'], r.get('lds_bytes_per_tile'), c.get('rows_read_rank0')))
except Exception as e: print(sys.argv[2], 'rc', sys.argv[3], 'no line')
PY
  grep -v amdgpu.ids gpurun_out/r4v_$tag.err | tail -1 | cut -c1-200; }
run default
GK_GROUP_MAX=64 run g64
GK_GROUP_MAX=32 run g32
GK_GROUP_MAX=16 run g16
