#!/bin/bash
# round 6, visit az: the device fuzz campaign of visit ax stopped on an EngineError (device flagged a pair the renderer denies) without
# naming its seed; the CPU build passes seeds 9500..10099.  Find the seed, then: the same seed again (deterministic?), with the
# popcount kernel instead of the totals rows (GK_FUSED_TOTALS=0), and on the bytecode kernel (GK_JIT=0)
set -u
export TMPDIR=/tmp
timeout 900 python tools/scratch/device_fuzz_campaign.py ${FIRST:-9790} ${LAST:-9870} 2>&1 | grep -v amdgpu.ids | cut -c1-1500 > gpurun_out/r06az_pass1.log
tail -5 gpurun_out/r06az_pass1.log
for seed in $(grep -o "^=== seed [0-9]*" gpurun_out/r06az_pass1.log | awk '{print $3}' | head -3); do
  echo "--- seed $seed again"; timeout 300 python tools/scratch/device_fuzz_campaign.py $seed $seed 2>&1 | grep -v amdgpu.ids | cut -c1-600 | tail -3
  echo "--- seed $seed, GK_FUSED_TOTALS=0"; GK_FUSED_TOTALS=0 timeout 300 python tools/scratch/device_fuzz_campaign.py $seed $seed 2>&1 | grep -v amdgpu.ids | cut -c1-600 | tail -3
  echo "--- seed $seed, GK_JIT=0"; GK_JIT=0 timeout 300 python tools/scratch/device_fuzz_campaign.py $seed $seed 2>&1 | grep -v amdgpu.ids | cut -c1-600 | tail -3
done
