#!/bin/bash
# round 3, call A: GPU suite, default bench line (world-size-1 RCCL reduce inside the step), kernel trace of the same
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/r03_a
mkdir -p "$O"
cd "$R"
timeout 1500 python -m pytest tests -m gpu -x -q > "$O/pytest.log" 2>&1
echo "pytest rc=$?" | tee -a "$O/pytest.log"
tail -5 "$O/pytest.log"
PG_BENCH_VERBOSE=1 timeout 600 python bench.py > "$O/bench_default.json" 2> "$O/bench_default.err"
echo "bench rc=$?"
tail -c 3000 "$O/bench_default.json"
cd /tmp && export TMPDIR=/tmp
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d "$O/prof" -o bench -- python $R/bench.py --steps 5 --warmup 1 --no-cpu-baseline --sites-steps 0 --stream-batches 0 > "$O/bench_under_rocprof.json" 2> "$O/prof.err"
echo "rocprof rc=$?"
find "$O/prof" -name "*.csv" | head
