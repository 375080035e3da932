#!/bin/bash
# round 3, call AJ: the final kernel SOURCES (comments edited after call AF: the counter files carry a hash of the sources) -- whole GPU suite, kernel stats + timeline inputs, PMC traffic + SQ counters of the fill (the
# kernel sources changed: the counter files of call J no longer belong to this build), default bench line, two fresh salts and
# one stress seed, read-length probe, BAM -> genotypes probe
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/r03_aj
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
python tools/pmc_traffic.py gpurun_out/pmc_r03 gpurun_out/r03_aj/traffic_r03.json 200000 > /dev/null 2>&1
python tools/sq_summary.py gpurun_out/sq gpurun_out/r03_aj/r03_sq_counters.json 200000 > /dev/null 2>&1
cp gpurun_out/r03_aj/traffic_r03.json profiles/traffic_r03.json; cp gpurun_out/r03_aj/r03_sq_counters.json profiles/r03_sq_counters.json
timeout 600 python bench.py > "$O/bench_default.json" 2> "$O/bench_default.err"
echo "bench rc=$? lines=$(wc -l < $O/bench_default.json)"
python -c "
import json; d=json.loads(open('$O/bench_default.json').readline()); r=d['roofline']; print(d['value'], d['ms_per_step'], r['frac'], r['hbm_measured_frac'], r['valu'].get('issue_frac'), r['traffic_source']['usable'], d['sites']['sites_per_s'], d['verified'], d['dist']['collective_ab']['with_vs_without'])"
