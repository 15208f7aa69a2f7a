#!/bin/bash
# Round-3 visit q: RCCL path at world size 1, four enqueues per pass (sweep, tail kernel, all-gather, totals kernel); GK_SHARD_GRAPH=1 against direct
set -u
tag=${1:-r3q}
mkdir -p gpurun_out; export TMPDIR=/tmp
GK_JIT_STRICT=1 timeout 600 python -m pytest tests/test_sweep_dist.py -m gpu -x -q 2>&1 | tail -15 > gpurun_out/${tag}_pytest_rccl.log; GK_SHARD_GRAPH=1 GK_JIT_STRICT=1 timeout 600 python -m pytest tests/test_sweep_dist.py -m gpu -x -q 2>&1 | tail -5 >> gpurun_out/${tag}_pytest_rccl.log
GK_SHARD_GRAPH=1 GK_FORCE_DIST=1 timeout 300 python bench.py --steps 200 --warmup 10 --lean > gpurun_out/${tag}_bench_rccl_world1_graph.json 2> gpurun_out/${tag}_bench_rccl_world1_graph.err
GK_SHARD_GRAPH=0 GK_FORCE_DIST=1 timeout 300 python bench.py --steps 200 --warmup 10 --lean > gpurun_out/${tag}_bench_rccl_world1_direct.json 2> gpurun_out/${tag}_bench_rccl_world1_direct.err
(cd /tmp && GK_FORCE_DIST=1 timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/${tag}_stats -o stats -- python $GRAFT_REPO_ROOT/bench.py --steps 50 --warmup 5 --lean > /dev/null 2> $GRAFT_REPO_ROOT/gpurun_out/${tag}_stats.err)
tail -6 gpurun_out/${tag}_pytest_rccl.log
for f in graph direct; do python - "$tag" "$f" <<'PY'
import sys, json
for line in open('gpurun_out/%s_bench_rccl_world1_%s.json' % (sys.argv[1], sys.argv[2])):
    if line.startswith('{'):
        d = json.loads(line)
        print('%s value %.4g ms_per_step %.4f kernel_ms %.4f pairs %d' % (sys.argv[2], d['value'], d['ms_per_step'], d['roofline']['avg_kernel_ms'], d['config']['global_violating_pairs']))
PY
done
find gpurun_out/${tag}_stats -name '*kernel_stats.csv' | head -1 | xargs -r head -8
