#!/bin/bash
# round 6, visit bo: GK_BIT (opaque copy + mask in front of every test of a formula value) -- the two open seeds, the gpu suite, smoke,
# PMC passes + rocprofv3 stats of the new text, lean 1 M / 10 M
set -u
export TMPDIR=/tmp
GK_PROBE_FIRST_ONLY=1 timeout 250 python tools/scratch/device_fuzz_probe.py 9820 K8sFuzz15 K8sFuzz19 gpu 2>&1 | grep -v amdgpu.ids | tail -1 | cut -c1-330
GK_PROBE_FIRST_ONLY=1 timeout 250 python tools/scratch/device_fuzz_probe.py 9833 K8sFuzz37 K8sFuzz38 gpu 2>&1 | grep -v amdgpu.ids | tail -1 | cut -c1-430
timeout 600 python tools/scratch/device_fuzz_campaign.py 9815 9840 2>&1 | grep -v amdgpu.ids | tail -3 | cut -c1-300
bash tools/gpu_visit.sh r06bo tests smoke pmc stats lean
timeout 900 python bench.py --config 2 --lean --steps 50 --warmup 5 --reviews 10000000 2>/dev/null | python -c "
import json,sys
j=json.loads(sys.stdin.read().strip().split('\n')[-1]); r=j['roofline']; print('lean 10M: step %.4f ms kernel %.4f ms frac %.4f' % (j['ms_per_step'], r['avg_kernel_ms'], r['frac']))"
