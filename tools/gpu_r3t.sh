#!/bin/bash
# Round-3 visit t: count kernels without clearing passes (one 1024-thread block per bitmap row, sums stored): the plain sweep, the
# sharded sweep at world size 1, the corpus sweep; kernel trace of the plain sweep
set -u
tag=${1:-r3t}
mkdir -p gpurun_out; export TMPDIR=/tmp
GK_JIT_STRICT=1 timeout 600 python -m pytest tests/test_sweep_dist.py tests/test_parity.py -m gpu -x -q -k "rccl or audit or synthetic or edge" 2>&1 | tail -4 > gpurun_out/${tag}_pytest.log
timeout 300 python bench.py --steps 200 --warmup 10 --lean > gpurun_out/${tag}_bench_config2.json 2> gpurun_out/${tag}_bench_config2.err
GK_FORCE_DIST=1 timeout 300 python bench.py --steps 200 --warmup 10 --lean > gpurun_out/${tag}_bench_rccl_world1.json 2> gpurun_out/${tag}_bench_rccl_world1.err
timeout 300 python bench.py --config 4 --steps 50 --warmup 5 --lean > gpurun_out/${tag}_bench_config4.json 2> gpurun_out/${tag}_bench_config4.err
(cd /tmp && timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/${tag}_stats -o stats -- python $GRAFT_REPO_ROOT/bench.py --steps 50 --warmup 5 --lean > /dev/null 2> $GRAFT_REPO_ROOT/gpurun_out/${tag}_stats.err)
tail -3 gpurun_out/${tag}_pytest.log
for f in config2 rccl_world1 config4; do python - "$tag" "$f" <<'PY'
import sys, json
for line in open('gpurun_out/%s_bench_%s.json' % (sys.argv[1], sys.argv[2])):
    if line.startswith('{'):
        d = json.loads(line)
        print('%s value %.4g ms_per_step %.4f kernel_ms(sum) %.4f frac %.4f' % (sys.argv[2], d['value'], d['ms_per_step'], d['roofline']['avg_kernel_ms'], d['roofline']['frac']))
PY
done
find gpurun_out/${tag}_stats -name '*kernel_stats.csv' | head -1 | xargs -r head -6
