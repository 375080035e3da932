#!/bin/bash
# round 3, call AP: the bench's default workspace is 128 GiB now (3 fill launches per step): kernel stats under rocprofv3 and the
# default bench line of that command (the counter files are per read and stay as collected)
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/r03_ap
mkdir -p "$O"
cd /tmp && export TMPDIR=/tmp
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d "$O/prof" -o bench -- python $R/bench.py --steps 5 --warmup 1 --no-cpu-baseline --sites-steps 0 --stream-batches 0 > "$O/bench_under_rocprof.json" 2> "$O/prof.err"
echo "rocprof rc=$?"
head -3 $O/prof/bench_kernel_stats.csv | cut -c1-120
cd "$R"
timeout 300 python bench.py > "$O/bench_default.json" 2> "$O/bench_default.err"
echo "bench rc=$? lines=$(wc -l < $O/bench_default.json)"
python -c "
import json; d=json.loads(open('$O/bench_default.json').readline()); r=d['roofline']; print(d['value'], d['ms_per_step'], r['launches'], r['avg_launch_ms'], r['frac'], r['hbm_measured_frac'], r['valu'].get('issue_frac'), r['traffic_source']['usable'], d['sites']['sites_per_s'], d['verified'], d['dist']['collective_ab']['with_vs_without'])"
