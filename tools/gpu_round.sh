#!/bin/bash
# One GPU visit: gpu parity tests, smoke, bench (+ phase-isolation timings), rocprofv3 kernel trace.  Writes gpurun_out/.
set -u
tag=${1:-run}
mkdir -p gpurun_out
python -m pytest tests -m gpu -x -q 2>&1 | tail -15 > gpurun_out/${tag}_pytest_gpu.log
python -c "import __graft_entry__ as g; g.smoke()" > gpurun_out/${tag}_smoke.log 2>&1
python bench.py --steps 50 --warmup 5 > gpurun_out/${tag}_bench.json 2> gpurun_out/${tag}_bench.err
for ph in 1 2 3; do
  GK_DBG_PHASE=$ph python bench.py --steps 30 --warmup 3 --no-cpu-baseline 2>/dev/null | python -c "
import sys, json
d = json.loads(sys.stdin.read().strip().splitlines()[-1])
print('GK_DBG_PHASE=$ph avg_kernel_ms', d['roofline']['avg_kernel_ms'])" >> gpurun_out/${tag}_phases.log 2>&1
done
export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/${tag}_prof -o ${tag} -- python bench.py --steps 30 --warmup 3 --no-cpu-baseline > gpurun_out/${tag}_prof_bench.json 2> gpurun_out/${tag}_prof.err
find gpurun_out/${tag}_prof -name '*kernel_stats*' | head -3
cat gpurun_out/${tag}_pytest_gpu.log | tail -3
cat gpurun_out/${tag}_smoke.log | tail -2
cat gpurun_out/${tag}_bench.json | tail -1
cat gpurun_out/${tag}_phases.log
