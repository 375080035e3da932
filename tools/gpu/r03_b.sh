#!/bin/bash
# round 3, call B: count / copy streams on their own hardware queues (priority level): suite, bench, kernel trace, PMC + SQ passes
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/r03_b
mkdir -p "$O"
cd "$R"
timeout 1500 python -m pytest tests -m gpu -x -q > "$O/pytest.log" 2>&1
echo "pytest rc=$?"; tail -3 "$O/pytest.log"
PG_BENCH_VERBOSE=1 timeout 600 python bench.py > "$O/bench_default.json" 2> "$O/bench_default.err"
echo "bench rc=$? lines=$(wc -l < $O/bench_default.json)"
PG_STREAM_PRIORITY=0 timeout 300 python bench.py --no-cpu-baseline --sites-steps 0 --stream-batches 0 --steps 5 > "$O/bench_prio0.json" 2> /dev/null
cd /tmp && export TMPDIR=/tmp
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d "$O/prof" -o bench -- python $R/bench.py --steps 5 --warmup 1 --no-cpu-baseline --sites-steps 0 --stream-batches 0 > "$O/bench_under_rocprof.json" 2> "$O/prof.err"
echo "rocprof rc=$?"
cd "$R"
PG_HEAD=${PG_HEAD:-unknown} bash tools/pmc_collect.sh r03 > "$O/pmc.log" 2>&1
PG_HEAD=${PG_HEAD:-unknown} bash tools/sq_collect.sh > "$O/sq.log" 2>&1
python - <<'PY'
import json
for f in ("bench_default", "bench_prio0", "bench_under_rocprof"):
    try:
        d = json.loads(open("gpurun_out/r03_b/%s.json" % f).readline())
        print(f, round(d["value"] / 1e6, 2), "M reads/s", round(d["ms_per_step"], 2), "ms/step, fill", round(d["roofline"]["avg_launch_ms"], 3), "ms", d["dist"].get("collective_ab", {}).get("with_vs_without"))
    except Exception as e:
        print(f, "failed", e)
PY
