# visit r05aa: the roofline clock (one event pair around consecutive launches of the kernel alone) against the pair per launch, and the collecting call
set -u; mkdir -p gpurun_out; export TMPDIR=/tmp
for k in 20 50 20; do
timeout 300 python bench.py --lean --steps $k --warmup 5 > gpurun_out/r05aa_s$k.json 2> gpurun_out/r05aa_s$k.err
python - gpurun_out/r05aa_s$k.json $k <<'PY'
import json, sys
j = json.loads(open(sys.argv[1]).read().strip().split('\n')[-1]); r = j['roofline']
print('steps %s: step %.4f ms | kernel (one pair, kernel alone) %.4f ms frac %.4f | pair per launch %.4f ms | whole step back to back %.4f ms | %s' % (sys.argv[2], j['ms_per_step'], r['avg_kernel_ms'], r['frac'], r['avg_kernel_ms_event_pair_per_launch'], r['avg_launch_ms_back_to_back'], r['clock']))
PY
done
timeout 300 python bench.py --config 1 --lean --steps 50 --warmup 5 2> /dev/null | python -c "
import sys, json
j = json.loads(sys.stdin.read().strip().split('\n')[-1]); r = j['roofline']; print('configs[1]: step %.4f kernel %.4f frac %.4f pair per launch %.4f' % (j['ms_per_step'], r['avg_kernel_ms'], r['frac'], r['avg_kernel_ms_event_pair_per_launch']))"
GK_JIT_STRICT=1 timeout 300 python -m pytest tests/test_parity.py -m gpu -x -q -k kernel_only 2>&1 | tail -2
