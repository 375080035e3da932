#!/bin/bash
# round 6, the tree with the lean gssw stage: everything profiles/r06_* is made from, on ONE box (as tools/gpu/r06_measure.sh did for
# the plain stage): stage counters of the other kernels, PMC traffic and SQ counters of the lean stage's two fill kernels together
# (separate rocprofv3 passes, --kernel-trace only), kernel stats of the bench command, the stage probe
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}; cd "$R"; O=$R/gpurun_out/r6meas2; mkdir -p $O
export TMPDIR=/tmp
PG_HEAD=r06-lean bash tools/pmc_collect.sh r06 > $O/pmc.log 2>&1; echo "pmc rc=$?"
PG_HEAD=r06-lean bash tools/sq_collect.sh > $O/sq.log 2>&1; echo "sq rc=$?"
python tools/pmc_traffic.py gpurun_out/pmc_r06 $O/traffic_r06.json > /dev/null 2> $O/traffic.err; echo "traffic rc=$?"; tail -2 $O/traffic.err
python tools/sq_summary.py gpurun_out/sq $O/r06_sq_counters.json 200000 "pg_fill_lean" 518 1.5 > /dev/null 2> $O/sqsum.err; echo "sqsum rc=$?"; tail -2 $O/sqsum.err
bash tools/stage_counters.sh 200000 > $O/stage_counters.log 2>&1; echo "stage counters rc=$?"
python tools/stage_counters_summary.py gpurun_out/stage_counters $O/r06_stage_counters.json 200000 > $O/stage_counters_summary.txt 2> $O/stage_counters_summary.err; echo "summary rc=$?"; cat $O/stage_counters_summary.txt
cd /tmp
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d "$O/prof" -o bench -- python $R/bench.py --steps 5 --warmup 1 --no-cpu-baseline --sites-steps 0 --stream-batches 0 --e2e-steps 0 --config5-graphs 0 --exact-shortcut-steps 0 --plain-steps 0 > "$O/bench_under_rocprof.json" 2> "$O/prof.err"; echo "rocprof rc=$?"
head -8 $O/prof/bench_kernel_stats.csv | cut -c1-200
cd "$R"
python tools/stage_probe.py 1000000 > $O/stage_probe.json 2> $O/stage_probe.err; echo "probe rc=$?"; tail -c 1500 $O/stage_probe.json
