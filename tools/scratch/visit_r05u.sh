# visit r05u (one box; r05t measured the same with a missing wait state in the writelane text: wrong, varying pair counts): result words kept in the wave (v_writelane, one LDS store per kind and part) and rolling LDS reads of scopes read
# at use, each against its A/B switch (GK_JIT_RES_LANES=0, GK_JIT_ROLL=0), on configs[2], [1] and the corpus; phase clocks; parity legs.
set -u; mkdir -p gpurun_out; export TMPDIR=/tmp
run() { tag=$1; shift; timeout 300 python bench.py "$@" --lean --steps 50 --warmup 5 > gpurun_out/r05u_$tag.json 2> gpurun_out/r05u_$tag.err; rc=$?
  python - gpurun_out/r05u_$tag.json $tag $rc <<'PY'
import json, sys
try:
    j = json.loads(open(sys.argv[1]).read().strip().split('\n')[-1]); r = j['roofline']; c = j['config']
    print('%s: step %.4f ms kernel %.4f ms frac %.4f algo %d rows_read %s pairs %s' % (sys.argv[2], j['ms_per_step'], r['avg_kernel_ms'], r['frac'], r['algo_bytes_per_launch'], c.get('rows_read_rank0'), c.get('global_violating_pairs')))
except Exception as e: print(sys.argv[2], 'rc', sys.argv[3], 'no line', e)
PY
  grep "gkgpu prof" gpurun_out/r05u_$tag.err | tail -1 | cut -c1-260; grep -v "gkgpu prof\|amdgpu.ids" gpurun_out/r05u_$tag.err | tail -1 | cut -c1-200; }
run c2_base
GK_JIT_ROLL=0 run c2_noroll
GK_JIT_RES_LANES=0 run c2_nolanes
GK_JIT_ROLL=0 GK_JIT_RES_LANES=0 run c2_neither
run c2_base2
GK_JIT_ROLL=0 run c2_noroll2
GK_JIT_RES_LANES=0 run c2_nolanes2
GK_JIT_ROLL=0 GK_JIT_RES_LANES=0 run c2_neither2
GK_KERNEL_PROF=1 run c2_base_prof
GK_KERNEL_PROF=1 GK_JIT_ROLL=0 GK_JIT_RES_LANES=0 run c2_neither_prof
run c1_base --config 1
GK_JIT_ROLL=0 GK_JIT_RES_LANES=0 run c1_neither --config 1
run c4_base --config 4
GK_JIT_ROLL=0 run c4_noroll --config 4
GK_JIT_RES_LANES=0 run c4_nolanes --config 4
GK_JIT_ROLL=0 GK_JIT_RES_LANES=0 run c4_neither --config 4
timeout 600 python bench.py --no-other-configs --oracle-sample 16384 --steps 50 --warmup 5 > gpurun_out/r05u_benchq.json 2> gpurun_out/r05u_benchq.err
python - <<'PY'
import json
try:
    j = json.loads(open('gpurun_out/r05u_benchq.json').read().strip().split('\n')[-1]); r = j['roofline']
    print('benchq', r['frac'], j['ms_per_step'], j.get('parity_sample', {}).get('pairs_equal'), j.get('parity_python_oracle', {}).get('pairs_equal'), j.get('parity_messages_compiled_independent', {}).get('messages_equal'),
          j.get('audit_result_totals', {}).get('independent_compiled_checker', {}).get('equal'), j.get('audit_result_totals', {}).get('host_pass_over_every_pair', {}).get('equal'))
except Exception as e: print('benchq: no line', e)
PY
tail -2 gpurun_out/r05u_benchq.err | cut -c1-300
GK_JIT_STRICT=1 timeout 900 python -m pytest tests/test_parity.py tests/test_kernel_emu.py tests/test_result_totals.py tests/test_library_patterns.py tests/test_template_fuzz.py -m gpu -x -q 2>&1 | tail -3
