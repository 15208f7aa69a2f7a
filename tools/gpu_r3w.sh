#!/bin/bash
# Round-3 visit w: admission latency (native caller threads) and the streaming configs[4] record on the final build
set -u
tag=${1:-r3w}
mkdir -p gpurun_out; export TMPDIR=/tmp
timeout 240 python tools/latency_probe.py > gpurun_out/${tag}_latency.json 2> gpurun_out/${tag}_latency.err
timeout 240 python bench.py --config 4 --streaming --offered 1000000 --stream-batches 16 > gpurun_out/${tag}_stream_1M.json 2> gpurun_out/${tag}_stream_1M.err
timeout 240 python bench.py --config 4 --streaming --offered 0 --stream-batches 16 > gpurun_out/${tag}_stream_closed.json 2> gpurun_out/${tag}_stream_closed.err
python - "$tag" <<'PY'
import json, sys
tag = sys.argv[1]
for f in ('stream_1M', 'stream_closed'):
    try:
        for line in open('gpurun_out/%s_%s.json' % (tag, f)):
            if line.startswith('{'):
                d = json.loads(line); print(f, json.dumps(d['stream'])[:500])
    except Exception as e: print(f, 'ERR', e)
try:
    d = json.load(open('gpurun_out/%s_latency.json' % tag))
    for r in d.get('native', {}).get('runs', d.get('runs', [])):
        if r.get('threads') in (1, 64, 256) and r.get('workers') == 2: print('latency', {k: (round(v, 1) if isinstance(v, float) else v) for k, v in r.items() if k in ('threads', 'p50_us', 'p99_us', 'reviews_per_s', 'mean_batch', 'window_us', 'workers', 'max_batch')})
except Exception as e: print('latency ERR', e)
PY
