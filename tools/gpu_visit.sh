#!/bin/bash
# One gpurun visit on an MI355X box.  usage: tools/gpu_visit.sh TAG STAGE...   stages (run in the order given):
#   tests      the whole gpu-marked suite (GK_JIT_STRICT=1: a hiprtc failure fails the test)
#   smoke      __graft_entry__.smoke()
#   bench      the default bench line (configs[2] + other_configs) as the driver runs it (--steps 20 --warmup 5)
#   benchq     the headline workload alone with its parity legs (independent compiled checker, python oracle on 16 384, RESULT totals)
#   lean       bench.py --lean --steps 50 (the headline kernel only: tuning runs)
#   stats      rocprofv3 --kernel-trace --stats of the lean command
#   pmc        rocprofv3 --pmc passes of the lean command (SQ, FETCH_SIZE, WRITE_SIZE: separate passes, --kernel-trace only)
#   pmc4       the SQ and FETCH_SIZE passes on bench.py --config 4 --lean (the 200-template corpus kernel)
#   c1 | c4    bench.py --config 1 | 4 --lean
#   stream     bench.py --config 4 --streaming (offered 1 M/s) and closed loop
#   env:X=Y    export X=Y for the following stages;  unset:X
# Everything lands under gpurun_out/TAG_*; copy what is to be judged into profiles/.
set -u
tag=$1; shift
mkdir -p gpurun_out; export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-$PWD}
pmc_summary() {
  for d in "$@"; do
    f=$(find gpurun_out/${tag}_pmc_$d -name '*counter_collection.csv' | head -1)
    [ -n "$f" ] && python - "$f" <<'PY'
import csv, sys, collections
acc = collections.defaultdict(lambda: collections.defaultdict(float)); n = collections.Counter()
for r in csv.DictReader(open(sys.argv[1])):
    k = r.get('Kernel_Name', '')
    if 'tiles' not in k: continue
    acc[k][r['Counter_Name']] += float(r['Counter_Value']); n[(k, r['Counter_Name'])] += 1
for k in acc:
    for c, v in sorted(acc[k].items()):
        print('%s %s per_dispatch=%.1f dispatches=%d' % (k[:20], c, v / n[(k, c)], n[(k, c)]))
PY
  done
}
for stage in "$@"; do
  case $stage in
    env:*) export "${stage#env:}";;
    unset:*) unset "${stage#unset:}";;
    tests) GK_JIT_STRICT=1 timeout 1500 python -m pytest tests -m gpu -x -q 2>&1 | tail -8 > gpurun_out/${tag}_pytest_gpu.log; tail -3 gpurun_out/${tag}_pytest_gpu.log;;
    smoke) timeout 300 python -c "import __graft_entry__ as g; g.smoke()" > gpurun_out/${tag}_smoke.log 2>&1; tail -2 gpurun_out/${tag}_smoke.log;;
    bench) s=$(date +%s); timeout 900 python bench.py --gpus 1 --steps 20 --warmup 5 > gpurun_out/${tag}_bench.json 2> gpurun_out/${tag}_bench.err; echo "bench rc=$? wall=$(( $(date +%s) - s ))s"; tail -c 2400 gpurun_out/${tag}_bench.json; tail -3 gpurun_out/${tag}_bench.err;;
    lean) timeout 600 python bench.py --lean --steps 50 --warmup 5 > gpurun_out/${tag}_lean${GK_VARIANT:-}.json 2> gpurun_out/${tag}_lean${GK_VARIANT:-}.err; python - gpurun_out/${tag}_lean${GK_VARIANT:-}.json <<'PY'
import json, sys
try:
    j = json.loads(open(sys.argv[1]).read().strip().split('\n')[-1]); r = j['roofline']
    print('lean: step %.4f ms kernel %.4f ms frac %.4f algo %d' % (j['ms_per_step'], r['avg_kernel_ms'], r['frac'], r['algo_bytes_per_launch']))
except Exception as e: print('lean: no line', e)
PY
      grep "gkgpu prof" gpurun_out/${tag}_lean${GK_VARIANT:-}.err | tail -1; grep -v "gkgpu prof\|amdgpu.ids" gpurun_out/${tag}_lean${GK_VARIANT:-}.err | tail -2;;
    benchq) timeout 600 python bench.py --no-other-configs --oracle-sample 16384 --steps 50 --warmup 5 > gpurun_out/${tag}_benchq${GK_VARIANT:-}.json 2> gpurun_out/${tag}_benchq${GK_VARIANT:-}.err; python - gpurun_out/${tag}_benchq${GK_VARIANT:-}.json <<'PY'
import json, sys
try:
    j = json.loads(open(sys.argv[1]).read().strip().split('\n')[-1]); r = j['roofline']
    print('benchq: step %.4f ms kernel %.4f ms frac %.4f | cpu-loop parity %s (n=%s) python-oracle parity %s (n=%s) totals equal %s' % (j['ms_per_step'], r['avg_kernel_ms'], r['frac'],
          j.get('parity_sample', {}).get('pairs_equal'), j.get('parity_sample', {}).get('n'), j.get('parity_python_oracle', {}).get('pairs_equal'), j.get('parity_python_oracle', {}).get('n'),
          j.get('audit_result_totals', {}).get('host_pass_over_every_pair', {}).get('equal')))
except Exception as e: print('benchq: no line', e)
PY
      tail -2 gpurun_out/${tag}_benchq${GK_VARIANT:-}.err;;
    c1|c4) timeout 600 python bench.py --config ${stage#c} --lean --steps 50 --warmup 5 > gpurun_out/${tag}_${stage}.json 2> gpurun_out/${tag}_${stage}.err; tail -c 1500 gpurun_out/${tag}_${stage}.json; tail -2 gpurun_out/${tag}_${stage}.err;;
    stream) timeout 600 python bench.py --config 4 --streaming > gpurun_out/${tag}_stream_offered_1M.json 2> gpurun_out/${tag}_stream.err; tail -c 1200 gpurun_out/${tag}_stream_offered_1M.json
            timeout 600 python bench.py --config 4 --streaming --offered 0 > gpurun_out/${tag}_stream_closed_loop.json 2>> gpurun_out/${tag}_stream.err; tail -c 1200 gpurun_out/${tag}_stream_closed_loop.json;;
    stats) (cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/${tag}_stats -o stats -- python $R/bench.py --steps 50 --warmup 5 --lean > /dev/null 2> $R/gpurun_out/${tag}_stats.err)
           find gpurun_out/${tag}_stats -name '*kernel_stats.csv' | head -1 | xargs -r head -6;;
    pmc) run_pmc() { name=$1; shift; timeout 400 rocprofv3 --pmc "$@" --kernel-trace --output-format csv -d gpurun_out/${tag}_pmc_$name -o $name -- python bench.py --steps 5 --warmup 1 --lean > gpurun_out/${tag}_pmc_$name.json 2> gpurun_out/${tag}_pmc_$name.err; }
         run_pmc sq SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY
         run_pmc fetch FETCH_SIZE
         run_pmc write WRITE_SIZE
         pmc_summary sq fetch write | tee gpurun_out/${tag}_pmc_summary.txt
         python - gpurun_out/${tag}_pmc_summary.txt $tag > gpurun_out/${tag}_pmc_latest.json <<'PY'
import json, re, sys
v = {}
for line in open(sys.argv[1]):
    m = re.match(r"\S+ (\S+) per_dispatch=([0-9.]+)", line)
    if m and "jit_tiles" in line: v[m.group(1)] = float(m.group(2))
try:
    RL = json.loads(open("gpurun_out/%s_pmc_fetch.json" % sys.argv[2]).read().strip().split("\n")[-1])["roofline"]
    ALGO, KHASH = RL["algo_bytes_per_launch"], RL.get("kernel_text_hash")
except Exception: ALGO = KHASH = None
if "FETCH_SIZE" in v and "WRITE_SIZE" in v:
    print(json.dumps({"config": 2, "reviews": 1000000, "algo_bytes_per_launch": ALGO, "kernel_text_hash": KHASH, "sq": {k: x for k, x in v.items() if k.startswith("SQ_")}, "hbm_bytes_per_launch": int((2 * v["FETCH_SIZE"] + v["WRITE_SIZE"]) * 1024), "fetch_kb": v["FETCH_SIZE"], "write_kb": v["WRITE_SIZE"],
                      "source": "profiles/%s_summary.txt (rocprofv3 --pmc, separate passes of `bench.py --steps 5 --warmup 1 --lean` with --kernel-trace only; HBM bytes = 2 x FETCH_SIZE KB (gfx950 correction, MI355X_MICROARCH.md) + WRITE_SIZE KB, KB = 1024 B)" % (sys.argv[2][:3] + "_pmc_" + sys.argv[2][3:] + "_config2_1M")}, indent=1))
PY
         ;;
    pmc4) run_pmc4() { name=$1; shift; timeout 400 rocprofv3 --pmc "$@" --kernel-trace --output-format csv -d gpurun_out/${tag}_pmc_$name -o $name -- python bench.py --config 4 --steps 5 --warmup 1 --lean > /dev/null 2> gpurun_out/${tag}_pmc_$name.err; }
         run_pmc4 sq4 SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY
         run_pmc4 fetch4 FETCH_SIZE
         pmc_summary sq4 fetch4 | tee gpurun_out/${tag}_pmc4_summary.txt;;
    *) echo "unknown stage $stage";;
  esac
done
