#!/bin/bash
# Round-3 visit p: the engine's RCCL path at world size 1 with ONE collective per sweep and captured-graph replays
set -u
tag=${1:-r3p}
mkdir -p gpurun_out; export TMPDIR=/tmp
GK_JIT_STRICT=1 timeout 600 python -m pytest tests/test_sweep_dist.py -m gpu -x -q 2>&1 | tail -15 > gpurun_out/${tag}_pytest_rccl.log
GK_FORCE_DIST=1 timeout 300 python bench.py --steps 200 --warmup 10 --lean > gpurun_out/${tag}_bench_rccl_world1_graph.json 2> gpurun_out/${tag}_bench_rccl_world1_graph.err
GK_SHARD_GRAPH=0 GK_FORCE_DIST=1 timeout 300 python bench.py --steps 200 --warmup 10 --lean > gpurun_out/${tag}_bench_rccl_world1_direct.json 2> gpurun_out/${tag}_bench_rccl_world1_direct.err
(cd /tmp && GK_FORCE_DIST=1 timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/${tag}_stats -o stats -- python $GRAFT_REPO_ROOT/bench.py --steps 50 --warmup 5 --lean > /dev/null 2> $GRAFT_REPO_ROOT/gpurun_out/${tag}_stats.err)
tail -6 gpurun_out/${tag}_pytest_rccl.log
for f in graph direct; do tail -1 gpurun_out/${tag}_bench_rccl_world1_$f.json | python -c "
import sys, json
d = json.loads(sys.stdin.read().strip().splitlines()[-1])
print('$f value %.4g ms_per_step %.4f kernel_ms %.4f pairs %d' % (d['value'], d['ms_per_step'], d['roofline']['avg_kernel_ms'], d['config']['global_violating_pairs']))"; tail -2 gpurun_out/${tag}_bench_rccl_world1_$f.err; done
find gpurun_out/${tag}_stats -name '*kernel_stats.csv' | head -1 | xargs -r head -8
