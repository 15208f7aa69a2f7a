#!/bin/bash
# round 6, visit av: records of the build with the totals rows (no popcount kernel behind a sweep)
set -u
bash tools/gpu_visit.sh r06av tests smoke pmc pmc4 stats
GK_KERNEL_PROF=$PWD/gpurun_out/r06av_marks_1M.bin timeout 600 python bench.py --config 2 --lean --steps 3 --warmup 1 > /dev/null 2> gpurun_out/r06av_prof_1M.err
grep "gkgpu prof" gpurun_out/r06av_prof_1M.err | tail -1
for n in 1000000 10000000; do
timeout 900 python bench.py --config 2 --lean --steps 50 --warmup 5 --reviews $n > gpurun_out/r06av_lean_$n.json 2> gpurun_out/r06av_lean_$n.err
python - gpurun_out/r06av_lean_$n.json <<'PY'
import json, sys
j = json.loads(open(sys.argv[1]).read().strip().split('\n')[-1]); r = j['roofline']
print('lean: step %.4f ms kernel %.4f ms frac %.4f' % (j['ms_per_step'], r['avg_kernel_ms'], r['frac']))
PY
done
