#!/bin/bash
# round 5, call e: counters.  (1) the kernels besides fill / traceback (tools/stage_counters.sh -> r05_stage_counters.json);
# (2) PMC traffic + SQ counters of the fill / traceback on this tree's kernel sources (traffic_r05.json, r05_sq_counters.json);
# (3) kernel stats of the bench command; (4) headline A/B with one and two fill streams
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}; cd "$R"; O=$R/gpurun_out/r5e; mkdir -p $O
export TMPDIR=/tmp
bash tools/stage_counters.sh 200000 > $O/stage_counters.log 2>&1; echo "stage counters rc=$?"
python tools/stage_counters_summary.py gpurun_out/stage_counters $O/r05_stage_counters.json 200000 > $O/stage_counters_summary.txt 2> $O/stage_counters_summary.err; echo "summary rc=$?"; cat $O/stage_counters_summary.txt
PG_HEAD=r05 bash tools/pmc_collect.sh r05 > $O/pmc.log 2>&1; echo "pmc rc=$?"
PG_HEAD=r05 bash tools/sq_collect.sh > $O/sq.log 2>&1; echo "sq rc=$?"
python tools/pmc_traffic.py gpurun_out/pmc_r05 $O/traffic_r05.json > /dev/null 2> $O/traffic.err; echo "traffic rc=$?"
python tools/sq_summary.py gpurun_out/sq $O/r05_sq_counters.json > /dev/null 2> $O/sqsum.err; echo "sqsum rc=$?"
cd /tmp
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d "$O/prof" -o bench -- python $R/bench.py --steps 5 --warmup 1 --no-cpu-baseline --sites-steps 0 --stream-batches 0 --e2e-steps 0 > "$O/bench_under_rocprof.json" 2> "$O/prof.err"; echo "rocprof rc=$?"
head -5 $O/prof/bench_kernel_stats.csv | cut -c1-160
cd "$R"
for rep in 1 2; do for fs in 1 2; do
  PG_FILL_STREAMS=$fs python bench.py --steps 10 --warmup 2 --no-cpu-baseline --sites-steps 0 --stream-batches 0 --e2e-steps 0 2> /dev/null | python -c "
import json,sys
d=json.loads([l for l in sys.stdin if l.startswith('{')][-1])
print(json.dumps({'fill_streams': $fs, 'rep': $rep, 'value': d['value'], 'ms_per_step': d['ms_per_step'], 'launches': d['roofline']['launches'], 'avg_launch_ms': d['roofline']['avg_launch_ms']}))" | tee -a $O/headline_fill_streams_ab.jsonl
done; done
for ws in 128 165; do
  python bench.py --steps 10 --warmup 2 --no-cpu-baseline --sites-steps 0 --stream-batches 0 --e2e-steps 0 --workspace-gib $ws 2> /dev/null | python -c "
import json,sys
d=json.loads([l for l in sys.stdin if l.startswith('{')][-1])
print(json.dumps({'workspace_gib': $ws, 'value': d['value'], 'ms_per_step': d['ms_per_step'], 'launches': d['roofline']['launches'], 'avg_launch_ms': d['roofline']['avg_launch_ms']}))" | tee -a $O/headline_workspace_ab.jsonl
done
