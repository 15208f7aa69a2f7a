# visit r05y (one box): the 200-template stream (configs[4]) at the offered 10^6 reviews/s with batches of 65 536 (the bench line's), 32 768 and
# 16 384 reviews, and the closed loop; latency is counted from the arrival of a batch's last review to its downloaded answer
set -u; mkdir -p gpurun_out; export TMPDIR=/tmp
run() { tag=$1; shift; timeout 300 python bench.py --config 4 --streaming "$@" > gpurun_out/r05y_$tag.json 2> gpurun_out/r05y_$tag.err
  python - gpurun_out/r05y_$tag.json $tag <<'PY'
import json, sys
try:
    j = json.loads(open(sys.argv[1]).read().strip().split('\n')[-1]); s = j.get('stream') or j.get('config', {})
    print(sys.argv[2], 'value %.4g' % j['value'], {k: (round(v, 3) if isinstance(v, float) else v) for k, v in j.items() if k in ('ms_per_step',)}, json.dumps({k: v for k, v in j.get('latency', j.get('config', {})).items() if 'p50' in k or 'p99' in k or 'batch' == k or 'flatten' in k or 'device' in k})[:400])
except Exception as e: print(sys.argv[2], 'no line', e)
PY
  tail -c 900 gpurun_out/r05y_$tag.json; echo; grep -v amdgpu.ids gpurun_out/r05y_$tag.err | tail -1 | cut -c1-200; }
run b65536
run b32768 --batch 32768 --stream-batches 32
run b16384 --batch 16384 --stream-batches 64
run b8192 --batch 8192 --stream-batches 128
run closed65536 --offered 0
run closed16384 --offered 0 --batch 16384 --stream-batches 64
