#!/bin/bash
# One GPU visit: gpu parity tests, smoke, bench (configs[2] default + configs[1]), rocprofv3 kernel trace.  Writes gpurun_out/.
set -u
tag=${1:-run}
mkdir -p gpurun_out
(nproc; free -g | head -2; rocm-smi --showmeminfo vram 2>/dev/null | grep -i total | head -2) > gpurun_out/${tag}_box.log 2>&1
python -m pytest tests -m gpu -x -q 2>&1 | tail -15 > gpurun_out/${tag}_pytest_gpu.log
python -c "import __graft_entry__ as g; g.smoke()" > gpurun_out/${tag}_smoke.log 2>&1
python bench.py > gpurun_out/${tag}_bench.json 2> gpurun_out/${tag}_bench.err
python bench.py --config 1 --steps 200 > gpurun_out/${tag}_bench_cfg1.json 2> gpurun_out/${tag}_bench_cfg1.err
export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/${tag}_prof -o ${tag} -- python bench.py --steps 30 --warmup 3 --no-cpu-baseline > gpurun_out/${tag}_prof_bench.json 2> gpurun_out/${tag}_prof.err
find gpurun_out/${tag}_prof -name '*kernel_stats*' | head -3
cat gpurun_out/${tag}_box.log
cat gpurun_out/${tag}_pytest_gpu.log | tail -3
cat gpurun_out/${tag}_smoke.log | tail -2
cat gpurun_out/${tag}_bench.json | tail -1
cat gpurun_out/${tag}_bench_cfg1.json | tail -1
tail -3 gpurun_out/${tag}_bench.err
