#!/bin/bash
# round 4: everything profiles/r04_* is made from, on ONE box, for the kernel sources of the tree as it is:
#   the -m gpu suite; PMC traffic (tools/pmc_collect.sh -> traffic_r04.json) and SQ counters (tools/sq_collect.sh ->
#   r04_sq_counters.json), each in rocprofv3 passes of its own with --kernel-trace only; kernel stats of the bench command
#   (r04_kernel_stats.csv, r04_bench_under_rocprof.json); the default bench line (r04_bench_default.json)
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/r04_measure
mkdir -p "$O"
cd "$R"
export TMPDIR=/tmp
timeout 1500 python -m pytest tests -m gpu -x -q -p no:cacheprovider > "$O/pytest.log" 2>&1; echo "pytest rc=$? $(tail -1 $O/pytest.log)"
PG_HEAD=${PG_HEAD:-r04} bash tools/pmc_collect.sh r04 > "$O/pmc.log" 2>&1; echo "pmc rc=$?"
PG_HEAD=${PG_HEAD:-r04} bash tools/sq_collect.sh > "$O/sq.log" 2>&1; echo "sq rc=$?"
python tools/pmc_traffic.py gpurun_out/pmc_r04 "$O/traffic_r04.json" > /dev/null 2> "$O/traffic.err"; echo "traffic rc=$?"
python tools/sq_summary.py gpurun_out/sq "$O/r04_sq_counters.json" > /dev/null 2> "$O/sqsum.err"; echo "sqsum rc=$?"
cp "$O/traffic_r04.json" profiles/traffic_r04.json 2>/dev/null; cp "$O/r04_sq_counters.json" profiles/r04_sq_counters.json 2>/dev/null
cd /tmp
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d "$O/prof" -o bench -- python $R/bench.py --steps 5 --warmup 1 --no-cpu-baseline --sites-steps 0 --stream-batches 0 > "$O/bench_under_rocprof.json" 2> "$O/prof.err"; echo "rocprof rc=$?"
head -4 $O/prof/bench_kernel_stats.csv | cut -c1-140
cd "$R"
timeout 400 python bench.py > "$O/bench_default.json" 2> "$O/bench_default.err"; echo "bench rc=$? lines=$(wc -l < $O/bench_default.json)"
