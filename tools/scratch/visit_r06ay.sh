#!/bin/bash
# round 6, visit ay: the two levels of the 10 M-object kernel against the process's allocation history (tools/scratch/alloc_history_probe.py)
set -u
export TMPDIR=/tmp
for rep in 1 2 3; do
  timeout 300 python tools/scratch/alloc_history_probe.py fresh 2>&1 | grep -v amdgpu.ids | tail -1
  timeout 300 python tools/scratch/alloc_history_probe.py after 2>&1 | grep -v amdgpu.ids | tail -2
done
