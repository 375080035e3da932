#!/bin/bash
# round 3, call AF: the final kernels -- whole GPU suite, kernel stats + timeline inputs, PMC traffic + SQ counters of the fill (the
# kernel sources changed: the counter files of call J no longer belong to this build), default bench line, two fresh salts and
# one stress seed, read-length probe, BAM -> genotypes probe
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/r03_af
mkdir -p "$O"
cd "$R"
timeout 1800 python -m pytest tests -m gpu -q > "$O/pytest.log" 2>&1
echo "pytest rc=$?"; tail -3 "$O/pytest.log"
cd /tmp && export TMPDIR=/tmp
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d "$O/prof" -o bench -- python $R/bench.py --steps 5 --warmup 1 --no-cpu-baseline --sites-steps 0 --stream-batches 0 > "$O/bench_under_rocprof.json" 2> "$O/prof.err"
echo "rocprof rc=$?"
cd "$R"
PG_HEAD=${PG_HEAD:-unknown} bash tools/pmc_collect.sh r03 > "$O/pmc.log" 2>&1
PG_HEAD=${PG_HEAD:-unknown} bash tools/sq_collect.sh > "$O/sq.log" 2>&1
python tools/pmc_traffic.py gpurun_out/pmc_r03 gpurun_out/r03_af/traffic_r03.json 200000 > /dev/null 2>&1
python tools/sq_summary.py gpurun_out/sq gpurun_out/r03_af/r03_sq_counters.json 200000 > /dev/null 2>&1
cp gpurun_out/r03_af/traffic_r03.json profiles/traffic_r03.json; cp gpurun_out/r03_af/r03_sq_counters.json profiles/r03_sq_counters.json
timeout 600 python bench.py > "$O/bench_default.json" 2> "$O/bench_default.err"
echo "bench rc=$? lines=$(wc -l < $O/bench_default.json)"
python -c "
import json; d=json.loads(open('$O/bench_default.json').readline()); r=d['roofline']; print(d['value'], d['ms_per_step'], r['frac'], r['hbm_measured_frac'], r['valu'].get('issue_frac'), r['traffic_source']['usable'], d['sites']['sites_per_s'], d['verified'], d['dist']['collective_ab']['with_vs_without'])"
for salt in 1155 1266; do
  PG_SEED_SALT=$salt timeout 900 python -m pytest tests/test_gpu_klib.py tests/test_gpu_parity.py tests/test_gpu_counts.py tests/test_gpu_path.py tests/test_gpu_general.py -m gpu -q -p no:cacheprovider > $O/salt_$salt.log 2>&1
  echo "salt $salt rc=$? $(tail -1 $O/salt_$salt.log)"
done
timeout 900 python tests/stress_parity.py 2000 707 > $O/stress_707.log 2>&1; echo "stress 707 rc=$? $(tail -1 $O/stress_707.log)"
python tools/readlen_probe.py 200000 100,150,200,250,251,300,400,480 > $O/readlen.json 2> $O/readlen.err; echo "readlen rc=$?"
PG_E2E_DIR=tools/e2e/_data PG_E2E_REPS=6 timeout 600 tools/e2e/run.sh 10000 30 16 0 0 1 > $O/e2e.log 2>&1; echo "e2e rc=$?"; tail -1 $O/e2e.log | cut -c1-120
cp gpurun_out/e2e_probe.json $O/e2e_probe.json 2>/dev/null
