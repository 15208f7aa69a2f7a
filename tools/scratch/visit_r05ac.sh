# visit r05ac (one box): the stream legs over 256 batches of 16 384 reviews (offered 10^6/s, and closed loop), admission through gk_query
# (tools/latency_probe.py), and host ingest alone by thread count on the box's CPUs (tools/ingest_probe.py: the CPU build)
set -u; mkdir -p gpurun_out; export TMPDIR=/tmp
timeout 200 python bench.py --config 4 --streaming --stream-batches 256 > gpurun_out/r05ac_stream_offered_1M.json 2> gpurun_out/r05ac_stream.err; tail -c 700 gpurun_out/r05ac_stream_offered_1M.json | head -c 500; echo
timeout 200 python bench.py --config 4 --streaming --offered 0 --stream-batches 256 > gpurun_out/r05ac_stream_closed_loop.json 2>> gpurun_out/r05ac_stream.err; head -c 300 gpurun_out/r05ac_stream_closed_loop.json; echo
timeout 400 python tools/latency_probe.py > gpurun_out/r05ac_query_latency.json 2> gpurun_out/r05ac_query_latency.err; python - <<'PY'
import json
try:
    j = json.load(open('gpurun_out/r05ac_query_latency.json'))
    for r in j['native']['runs']:
        if r.get('threads') in (1, 256): print('native', {k: (round(v, 1) if isinstance(v, float) else v) for k, v in r.items()})
except Exception as e: print('latency: no record', e)
PY
for n in 16384 65536; do for th in 8 16 32 64; do echo "ingest $n reviews, $th threads:"; timeout 120 python tools/ingest_probe.py --reviews $n --threads $th --repeat 6 --config 4 2>&1 | tail -2; done; done > gpurun_out/r05ac_ingest_threads.log 2>&1; cat gpurun_out/r05ac_ingest_threads.log | cut -c1-160
