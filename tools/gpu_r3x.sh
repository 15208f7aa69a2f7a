#!/bin/bash
# Round-3 visit x: the whole gpu suite and smoke on the last build of the round
set -u
tag=${1:-r3x}
mkdir -p gpurun_out; export TMPDIR=/tmp
GK_JIT_STRICT=1 timeout 1000 python -m pytest tests -m gpu -q 2>&1 | tail -8 > gpurun_out/${tag}_pytest_gpu.log
timeout 200 python -c "import __graft_entry__ as g; g.smoke()" > gpurun_out/${tag}_smoke.log 2>&1
tail -4 gpurun_out/${tag}_pytest_gpu.log
tail -1 gpurun_out/${tag}_smoke.log
