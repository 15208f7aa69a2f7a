#!/bin/bash
# Round-3 visit j: RESULT totals with the device deciding which violating pairs can have more than one result (totals plans,
# counting loops F_ENDLOOP2, plan-local guards): the new tests on both GPU backends, the policy-compiler fuzz with its totals
# check, the default bench line (audit_result_totals now reports rendered pairs and the check against the host pass over every
# violating pair), the corpus
set -u
tag=${1:-r3j}
mkdir -p gpurun_out; export TMPDIR=/tmp
GK_JIT_STRICT=1 timeout 900 python -m pytest tests -m gpu -x -q -k "result_totals or audit or bench_legs or resident or template_fuzz or regressions" 2>&1 | tail -5 | tee gpurun_out/${tag}_pytest_gpu.log
timeout 900 python bench.py --no-cpu-baseline > gpurun_out/${tag}_bench_config2.json 2> gpurun_out/${tag}_bench_config2.err
timeout 600 python bench.py --config 4 --steps 50 --warmup 5 --no-cpu-baseline > gpurun_out/${tag}_bench_config4.json 2> gpurun_out/${tag}_bench_config4.err
python - "$tag" <<'PY'
import json, sys
tag = sys.argv[1]
for f in ('bench_config2', 'bench_config4'):
    try: d = json.loads(open('gpurun_out/%s_%s.json' % (tag, f)).read().strip().splitlines()[-1])
    except Exception as e: print(f, 'ERR', e); continue
    print(f, 'value %.4g ms_per_step %.4f frac %.4f kernel_ms %.4f' % (d['value'], d['ms_per_step'], d['roofline']['frac'], d['roofline']['avg_kernel_ms']))
    print(' totals', d.get('audit_result_totals'))
PY
tail -3 gpurun_out/${tag}_bench_config2.err gpurun_out/${tag}_bench_config4.err | grep -v amdgpu.ids
