# visit r05s (one box): kernel_body.inc with the next item's slots started from zeros (no wait behind the request, no spills) against
# the body of commit 2e1cb22 (tools/scratch/kernel_body_r05r.inc, through GK_JIT_BODY_FILE), one and two chunks in flight per wave,
# on configs[2], [1] and the corpus; per-phase clocks; then the parity legs of the headline workload on the new body.
set -u; mkdir -p gpurun_out; export TMPDIR=/tmp
OLD=$PWD/tools/scratch/kernel_body_r05r.inc   # (not kept in the tree: git show 2e1cb22:gatekeeper_amd/csrc/kernel_body.inc > tools/scratch/kernel_body_r05r.inc)
run() { tag=$1; shift; timeout 300 python bench.py "$@" --lean --steps 50 --warmup 5 > gpurun_out/r05s_$tag.json 2> gpurun_out/r05s_$tag.err; rc=$?
  python - gpurun_out/r05s_$tag.json $tag $rc <<'PY'
import json, sys
try:
    j = json.loads(open(sys.argv[1]).read().strip().split('\n')[-1]); r = j['roofline']; c = j['config']
    print('%s: step %.4f ms kernel %.4f ms frac %.4f algo %d rows_read %s pairs %s' % (sys.argv[2], j['ms_per_step'], r['avg_kernel_ms'], r['frac'], r['algo_bytes_per_launch'], c.get('rows_read_rank0'), c.get('global_violating_pairs')))
except Exception as e: print(sys.argv[2], 'rc', sys.argv[3], 'no line', e)
PY
  grep "gkgpu prof" gpurun_out/r05s_$tag.err | tail -1 | cut -c1-260; grep -v "gkgpu prof\|amdgpu.ids" gpurun_out/r05s_$tag.err | tail -1 | cut -c1-200; }
run c2_new
GK_JIT_BODY_FILE=$OLD run c2_old
GK_JIT_PREFETCH=2 run c2_new_p2
run c2_new2
GK_JIT_BODY_FILE=$OLD run c2_old2
GK_KERNEL_PROF=1 run c2_new_prof
GK_KERNEL_PROF=1 GK_JIT_PREFETCH=2 run c2_new_p2_prof
run c1_new --config 1
GK_JIT_BODY_FILE=$OLD run c1_old --config 1
GK_JIT_PREFETCH=2 run c1_new_p2 --config 1
run c4_new --config 4
GK_JIT_BODY_FILE=$OLD run c4_old --config 4
GK_JIT_PREFETCH=2 run c4_new_p2 --config 4
GK_STAGGER=40 run c2_new_st40
GK_JIT_PREFETCH=2 GK_STAGGER=40 run c2_new_p2_st40
# parity legs of the headline workload on the new body (independent compiled checker over every object, python oracle on 16 384, totals)
timeout 600 python bench.py --no-other-configs --oracle-sample 16384 --steps 50 --warmup 5 > gpurun_out/r05s_benchq.json 2> gpurun_out/r05s_benchq.err
python - <<'PY'
import json
try:
    j = json.loads(open('gpurun_out/r05s_benchq.json').read().strip().split('\n')[-1]); r = j['roofline']
    print('benchq', r['frac'], j['ms_per_step'], j.get('parity_sample', {}).get('pairs_equal'), j.get('parity_python_oracle', {}).get('pairs_equal'), j.get('parity_messages_compiled_independent', {}).get('messages_equal'),
          j.get('audit_result_totals', {}).get('independent_compiled_checker', {}).get('equal'), j.get('audit_result_totals', {}).get('host_pass_over_every_pair', {}).get('equal'))
except Exception as e: print('benchq: no line', e)
PY
tail -2 gpurun_out/r05s_benchq.err | cut -c1-300
GK_JIT_STRICT=1 timeout 600 python -m pytest tests/test_parity.py tests/test_kernel_emu.py tests/test_result_totals.py -m gpu -x -q 2>&1 | tail -3
