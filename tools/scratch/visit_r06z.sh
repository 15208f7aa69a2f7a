#!/bin/bash
# round 6, visit z: per-workgroup timeline of the build with a priority per phase (3003), 1 M and 10 M objects
set -u
mkdir -p gpurun_out; export TMPDIR=/tmp
GK_KERNEL_PROF=$PWD/gpurun_out/r06z_marks_1M.bin timeout 600 python bench.py --config 2 --lean --steps 3 --warmup 1 > /dev/null 2> gpurun_out/r06z_prof_1M.err
grep "gkgpu prof" gpurun_out/r06z_prof_1M.err | tail -1
GK_JIT_PRIO=0 GK_KERNEL_PROF=$PWD/gpurun_out/r06z_marks_1M_prio0.bin timeout 600 python bench.py --config 2 --lean --steps 3 --warmup 1 > /dev/null 2> gpurun_out/r06z_prof_1M_prio0.err
grep "gkgpu prof" gpurun_out/r06z_prof_1M_prio0.err | tail -1
GK_KERNEL_PROF=$PWD/gpurun_out/r06z_marks_3M.bin timeout 600 python bench.py --config 2 --lean --steps 3 --warmup 1 --reviews 3000000 > /dev/null 2> gpurun_out/r06z_prof_3M.err
grep "gkgpu prof" gpurun_out/r06z_prof_3M.err | tail -1
