#!/bin/bash
# round 6, visit al: the per-launch counters in pinned host memory (no 8-byte device-to-host copy in the collecting call) A/B, then the gpu suite
set -u
mkdir -p gpurun_out; export TMPDIR=/tmp
for rep in 1 2; do
echo "== counters in device memory"; GK_LC_PINNED=0 python tools/scratch/step_overhead_probe.py 2>/dev/null | grep -v "^steps  50\|^steps 100"
echo "== counters in pinned host memory"; python tools/scratch/step_overhead_probe.py 2>/dev/null | grep -v "^steps  50\|^steps 100"
done
bash tools/gpu_visit.sh r06al tests smoke
