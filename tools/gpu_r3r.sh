#!/bin/bash
# Round-3 visit r: byte-position masks for split()/trim() predicates (one mask per class and separator instead of a byte-wise scan
# per predicate): length-class parity tests on the device, the corpus sweep and its per-phase clocks, configs[2] unchanged?
set -u
tag=${1:-r3r}
mkdir -p gpurun_out; export TMPDIR=/tmp
GK_JIT_STRICT=1 timeout 900 python -m pytest tests/test_split_masks.py tests/test_sweep_dist.py tests/test_library_patterns.py -m gpu -x -q 2>&1 | tail -6 > gpurun_out/${tag}_pytest.log
timeout 300 python bench.py --config 4 --steps 50 --warmup 5 --lean > gpurun_out/${tag}_bench_config4.json 2> gpurun_out/${tag}_bench_config4.err
GK_KERNEL_PROF=1 timeout 300 python bench.py --config 4 --steps 10 --warmup 3 --lean 2>&1 >/dev/null | grep "gkgpu prof" | tail -4 > gpurun_out/${tag}_corpus_prof.log
timeout 300 python bench.py --steps 100 --warmup 5 --lean > gpurun_out/${tag}_bench_config2.json 2> gpurun_out/${tag}_bench_config2.err
GK_FORCE_DIST=1 timeout 300 python bench.py --steps 200 --warmup 10 --lean > gpurun_out/${tag}_bench_rccl_world1.json 2> gpurun_out/${tag}_bench_rccl_world1.err
tail -3 gpurun_out/${tag}_pytest.log
cat gpurun_out/${tag}_corpus_prof.log
for f in config4 config2 rccl_world1; do python - "$tag" "$f" <<'PY'
import sys, json
for line in open('gpurun_out/%s_bench_%s.json' % (sys.argv[1], sys.argv[2])):
    if line.startswith('{'):
        d = json.loads(line)
        print('%s value %.4g ms_per_step %.4f kernel_ms(sum) %.4f frac %.4f' % (sys.argv[2], d['value'], d['ms_per_step'], d['roofline']['avg_kernel_ms'], d['roofline']['frac']))
PY
done
