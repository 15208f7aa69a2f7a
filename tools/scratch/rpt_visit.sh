# configs[1] (30 x 100 k) at different row-group sizes: does a small table fill the GPU better with smaller groups?
set -u; mkdir -p gpurun_out
for r in 0 128 64; do
  if [ $r = 0 ]; then unset GK_RPT; else export GK_RPT=$r; fi
  timeout 300 python bench.py --config 1 --lean --steps 200 --warmup 20 > gpurun_out/r4s_c1_rpt$r.json 2> gpurun_out/r4s_c1_rpt$r.err
  python - gpurun_out/r4s_c1_rpt$r.json $r <<'PY'
import json, sys
j = json.loads(open(sys.argv[1]).read().strip().split('\n')[-1]); r = j['roofline']
print('rpt', sys.argv[2], 'step %.4f ms kernel %.4f ms frac %.3f lds %s' % (j['ms_per_step'], r['avg_kernel_ms'], r['frac'], r.get('lds_bytes_per_tile')))
PY
done
