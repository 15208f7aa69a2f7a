#!/bin/bash
# round 6, visit s: raw phase marks of the dominant kernel (GK_KERNEL_PROF=<file>: 8 clock samples per row group) for a per-workgroup timeline
set -u
mkdir -p gpurun_out; export TMPDIR=/tmp
GK_KERNEL_PROF=$PWD/gpurun_out/r06s_marks_1M.bin timeout 600 python bench.py --config 2 --lean --steps 3 --warmup 1 > gpurun_out/r06s_prof_1M.json 2> gpurun_out/r06s_prof_1M.err
grep "gkgpu prof" gpurun_out/r06s_prof_1M.err | tail -2
GK_KERNEL_PROF=$PWD/gpurun_out/r06s_marks_c1.bin timeout 600 python bench.py --config 1 --lean --steps 3 --warmup 1 > gpurun_out/r06s_prof_c1.json 2> gpurun_out/r06s_prof_c1.err
grep "gkgpu prof" gpurun_out/r06s_prof_c1.err | tail -2
timeout 600 python bench.py --config 2 --lean --steps 50 --warmup 5 > gpurun_out/r06s_lean.json 2> gpurun_out/r06s_lean.err
tail -c 600 gpurun_out/r06s_lean.json
ls -la gpurun_out/r06s_marks*
