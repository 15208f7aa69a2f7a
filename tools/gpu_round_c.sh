#!/bin/bash
# GPU visit: full gpu parity suite, smoke, default bench (configs[2]), rocprofv3 kernel stats + PMC passes of the same
# command, the sharded path through RCCL at world size 1, configs[1] and the 200-template corpus, admission latency.
set -u
tag=${1:-run}
mkdir -p gpurun_out; export TMPDIR=/tmp
timeout 900 python -m pytest tests -m gpu -x -q 2>&1 | tail -15 > gpurun_out/${tag}_pytest_gpu.log
timeout 300 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" > gpurun_out/${tag}_smoke.log 2>&1
timeout 600 python bench.py > gpurun_out/${tag}_bench.json 2> gpurun_out/${tag}_bench.err
(cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/${tag}_stats -o stats -- python $GRAFT_REPO_ROOT/bench.py --steps 50 --warmup 5 --no-cpu-baseline > /dev/null 2> $GRAFT_REPO_ROOT/gpurun_out/${tag}_stats.err)
GK_BENCH_ARGS="" timeout 900 bash tools/gpu_pmc.sh ${tag}pmc > gpurun_out/${tag}_pmc.log 2>&1
GK_FORCE_DIST=1 timeout 600 python bench.py --no-cpu-baseline --steps 100 > gpurun_out/${tag}_bench_rccl_w1.json 2> gpurun_out/${tag}_bench_rccl_w1.err
timeout 600 python bench.py --config 1 > gpurun_out/${tag}_bench_config1.json 2> gpurun_out/${tag}_bench_config1.err
GK_PLAN_TIMING=1 timeout 900 python bench.py --config 4 --steps 50 > gpurun_out/${tag}_bench_config4.json 2> gpurun_out/${tag}_bench_config4.err
timeout 600 python tools/latency_probe.py > gpurun_out/${tag}_latency.json 2> gpurun_out/${tag}_latency.err
tail -3 gpurun_out/${tag}_pytest_gpu.log
tail -2 gpurun_out/${tag}_smoke.log
for f in bench bench_rccl_w1 bench_config1 bench_config4; do echo "== $f"; tail -c 1500 gpurun_out/${tag}_$f.json; tail -3 gpurun_out/${tag}_$f.err; done
find gpurun_out/${tag}_stats -name '*kernel_stats.csv' | head -1 | xargs -r head -8
tail -30 gpurun_out/${tag}_pmc.log
tail -c 1200 gpurun_out/${tag}_latency.json
