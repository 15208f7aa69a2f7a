#!/bin/bash
# GPU visit: gpu parity tests, bench configs[2] (default), PMC passes of the same command.
set -u
tag=${1:-run}
mkdir -p gpurun_out
python -m pytest tests -m gpu -x -q 2>&1 | tail -15 > gpurun_out/${tag}_pytest_gpu.log
GK_PROFILE_HOST=1 python bench.py > gpurun_out/${tag}_bench.json 2> gpurun_out/${tag}_bench.err
GK_BENCH_ARGS="" bash tools/gpu_pmc.sh ${tag}pmc > gpurun_out/${tag}_pmc.log 2>&1
cat gpurun_out/${tag}_pytest_gpu.log | tail -3
cat gpurun_out/${tag}_bench.json | tail -1
grep gkgpu gpurun_out/${tag}_bench.err | tail -3
cat gpurun_out/${tag}_pmc.log | tail -40
