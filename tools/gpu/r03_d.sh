#!/bin/bash
# round 3, call D: where the config-3 leg loses against config 2 -- kernel stats of the sites workload
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/r03_d
mkdir -p "$O"
cd /tmp && export TMPDIR=/tmp
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d "$O/prof" -o c3 -- python $R/bench.py --workload config3 --steps 3 --warmup 1 --no-cpu-baseline > "$O/bench_config3.json" 2> "$O/prof.err"
echo "rc=$?"
cat "$O/prof/c3_kernel_stats.csv"
python - <<PY
import json
d = json.loads(open("$O/bench_config3.json").readline())
print(d["value"], d["ms_per_step"], d["roofline"]["avg_launch_ms"], d["roofline"]["launches"], d["kernel_ms"])
PY
