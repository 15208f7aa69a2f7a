set -u
mkdir -p gpurun_out
for th in 64 32; do
  GK_PROFILE_HOST=1 GK_HOST_THREADS=$th timeout 300 python bench.py --no-other-configs --no-cpu-baseline --oracle-sample 0 --steps 20 --warmup 5 > gpurun_out/r4o_th$th.json 2> gpurun_out/r4o_th$th.err
  grep "gkgpu host" gpurun_out/r4o_th$th.err | head -3
  python - gpurun_out/r4o_th$th.json <<'PY'
import json, sys
j = json.loads(open(sys.argv[1]).read().strip().split('\n')[-1]); e = j['end_to_end']
print('threads', e['host_threads'], 'first', e['flatten_s'], e['json_MBps'], 'second', e.get('second_table_of_the_same_batch'))
PY
done
