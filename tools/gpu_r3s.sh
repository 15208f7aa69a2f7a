#!/bin/bash
# Round-3 visit s: whole gpu suite at the K8sUniqueLabel / split-mask / regex-word commit; corpus sweep with the word-wise regex DFA
set -u
tag=${1:-r3s}
mkdir -p gpurun_out; export TMPDIR=/tmp
GK_JIT_STRICT=1 timeout 1200 python -m pytest tests -m gpu -q 2>&1 | tail -8 > gpurun_out/${tag}_pytest_gpu.log
timeout 300 python bench.py --config 4 --steps 50 --warmup 5 --lean > gpurun_out/${tag}_bench_config4.json 2> gpurun_out/${tag}_bench_config4.err
GK_KERNEL_PROF=1 timeout 300 python bench.py --config 4 --steps 10 --warmup 3 --lean 2>&1 >/dev/null | grep "gkgpu prof" | tail -4 > gpurun_out/${tag}_corpus_prof.log
tail -4 gpurun_out/${tag}_pytest_gpu.log
cat gpurun_out/${tag}_corpus_prof.log
python - "$tag" <<'PY'
import sys, json
for line in open('gpurun_out/%s_bench_config4.json' % sys.argv[1]):
    if line.startswith('{'):
        d = json.loads(line)
        print('config4 value %.4g ms_per_step %.4f kernel_ms(sum) %.4f' % (d['value'], d['ms_per_step'], d['roofline']['avg_kernel_ms']))
PY
