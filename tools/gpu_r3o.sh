#!/bin/bash
# Round-3 visit o: the whole gpu suite on the final build; per-phase clocks of the corpus sweep (GK_KERNEL_PROF) per plan group
set -u
tag=${1:-r3o}
mkdir -p gpurun_out; export TMPDIR=/tmp
GK_JIT_STRICT=1 timeout 1200 python -m pytest tests -m gpu -q 2>&1 | tail -8 > gpurun_out/${tag}_pytest_gpu.log
GK_KERNEL_PROF=1 timeout 300 python bench.py --config 4 --steps 10 --warmup 3 --lean 2>&1 >/dev/null | grep "gkgpu prof" | tail -8 > gpurun_out/${tag}_corpus_prof.log
timeout 300 python bench.py --config 4 --steps 50 --warmup 5 --lean > gpurun_out/${tag}_bench_config4.json 2> gpurun_out/${tag}_bench_config4.err
tail -4 gpurun_out/${tag}_pytest_gpu.log
cat gpurun_out/${tag}_corpus_prof.log
tail -1 gpurun_out/${tag}_bench_config4.json | cut -c1-600
