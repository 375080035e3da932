#!/bin/bash
# round 6: PMC traffic and SQ counters of the PLAIN gssw stage's fill kernel (PG_LEAN=0) on the same kernel sources as tools/gpu/r06_measure2.sh
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}; cd "$R"; O=$R/gpurun_out/r6meas3; mkdir -p $O
export TMPDIR=/tmp PG_LEAN=0
rm -rf gpurun_out/pmc_r06plain gpurun_out/sq
PG_HEAD=r06-lean-tree-plain-stage bash tools/pmc_collect.sh r06plain > $O/pmc.log 2>&1; echo "pmc rc=$?"
PG_HEAD=r06-lean-tree-plain-stage bash tools/sq_collect.sh > $O/sq.log 2>&1; echo "sq rc=$?"
python tools/pmc_traffic.py gpurun_out/pmc_r06plain $O/traffic_r06_plain_stage.json > /dev/null 2> $O/traffic.err; echo "traffic rc=$?"; tail -2 $O/traffic.err
python tools/sq_summary.py gpurun_out/sq $O/r06_sq_counters_plain_stage.json > /dev/null 2> $O/sqsum.err; echo "sqsum rc=$?"; tail -2 $O/sqsum.err
