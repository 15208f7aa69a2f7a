#!/bin/bash
# Round-3 visit v: overlapped exchange (two slot buffers + exchange stream; the default at world size > 1) exercised at world size 1
set -u
tag=${1:-r3v}
mkdir -p gpurun_out; export TMPDIR=/tmp
GK_JIT_STRICT=1 timeout 600 python -m pytest tests/test_sweep_dist.py -m gpu -q 2>&1 | tail -6 > gpurun_out/${tag}_pytest_rccl.log
GK_SHARD_OVERLAP=1 GK_FORCE_DIST=1 timeout 300 python bench.py --steps 200 --warmup 10 --lean > gpurun_out/${tag}_bench_rccl_world1_overlap.json 2> gpurun_out/${tag}_bench_rccl_world1_overlap.err
GK_SHARD_OVERLAP=0 GK_FORCE_DIST=1 timeout 300 python bench.py --steps 200 --warmup 10 --lean > gpurun_out/${tag}_bench_rccl_world1_serial.json 2> gpurun_out/${tag}_bench_rccl_world1_serial.err
tail -4 gpurun_out/${tag}_pytest_rccl.log
for f in overlap serial; do python - "$tag" "$f" <<'PY'
import sys, json
for line in open('gpurun_out/%s_bench_rccl_world1_%s.json' % (sys.argv[1], sys.argv[2])):
    if line.startswith('{'):
        d = json.loads(line)
        print('%s value %.4g ms_per_step %.4f kernel_ms %.4f pairs %d' % (sys.argv[2], d['value'], d['ms_per_step'], d['roofline']['avg_kernel_ms'], d['config']['global_violating_pairs']))
PY
tail -2 gpurun_out/${tag}_bench_rccl_world1_$f.err | grep -v "amdgpu.ids\|socket.cpp"
done
