#!/bin/bash
# round 4, call D: the tree after the integer additions, equal chunks, non-temporal walk loads and the N > 1 hardening of bench.py:
# whole -m gpu suite (with the 8-rank shared-device test), counters on the new kernel sources (HBM traffic, SQ), kernel stats of
# the default bench command, the default bench line
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/r04_d
mkdir -p "$O"
cd "$R"
export TMPDIR=/tmp
timeout 1500 python -m pytest tests -m gpu -x -q -p no:cacheprovider > "$O/pytest.log" 2>&1; echo "pytest rc=$? $(tail -1 $O/pytest.log)"
PG_HEAD=r04_d bash tools/pmc_collect.sh r04 > "$O/pmc.log" 2>&1; echo "pmc rc=$?"
PG_HEAD=r04_d bash tools/sq_collect.sh > "$O/sq.log" 2>&1; echo "sq rc=$?"
python tools/pmc_traffic.py gpurun_out/pmc_r04 "$O/traffic_r04.json" > /dev/null 2> "$O/traffic.err"; echo "traffic rc=$?"
python tools/sq_summary.py gpurun_out/sq "$O/r04_sq_counters.json" > /dev/null 2> "$O/sqsum.err"; echo "sqsum rc=$?"
cp "$O/traffic_r04.json" profiles/traffic_r04.json 2>/dev/null; cp "$O/r04_sq_counters.json" profiles/r04_sq_counters.json 2>/dev/null
cd /tmp
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d "$O/prof" -o bench -- python $R/bench.py --steps 5 --warmup 1 --no-cpu-baseline --sites-steps 0 --stream-batches 0 > "$O/bench_under_rocprof.json" 2> "$O/prof.err"; echo "rocprof rc=$?"
head -4 $O/prof/bench_kernel_stats.csv | cut -c1-140
cd "$R"
timeout 400 python bench.py > "$O/bench_default.json" 2> "$O/bench_default.err"; echo "bench rc=$? lines=$(wc -l < $O/bench_default.json)"
python - <<PY
import json
d = json.loads(open("$O/bench_default.json").readline()); r = d["roofline"]
print(d["value"], d["ms_per_step"], r["launches"], r["avg_launch_ms"], r["frac"], r["hbm_measured_frac"], r["valu"].get("issue_frac"), r["valu"].get("issue_frac_all_packed"), d["sites"]["sites_per_s"], d["verified"], d["sites"].get("verified"), d["dist"]["collective_ab"]["with_vs_without"])
PY
