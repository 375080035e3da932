#!/bin/bash
# workflow timeline: e2e probe with PG_WORKFLOW_TRACE (last run of the process) + HIP API trace of the same
R="$GRAFT_REPO_ROOT"; cd "$R" || exit 1
O=$R/gpurun_out/n; mkdir -p $O; rm -f $O/*.csv
export PG_E2E_DIR=$R/tools/e2e/_data
W=$PG_E2E_DIR
export PG_E2E_REPS=3
export PG_WORKFLOW_TRACE=$O/trace.tsv
( time bash tools/e2e/run.sh 10000 30 ${1:-32} 512 ${2:-8} 1 ) > $O/e2e_run.log 2>&1; tail -5 $O/e2e_run.log | cut -c1-300
cp gpurun_out/e2e_probe.json $O/e2e_probe.json
export TMPDIR=/tmp
export PG_WORKFLOW_TRACE=$O/trace_prof.tsv
timeout 300 rocprofv3 --hip-runtime-trace --kernel-trace --output-format csv -d $O/hip -o e2e -- $W/grmpy_batch $W/ref.fa $W/manifest.txt $W/graphs.txt ${1:-32} $W/g5.json 512 ${2:-8} 1 > $O/hip.json 2> $O/hip.err
cat $O/hip.json | cut -c1-1200
find $O/hip -name "*.csv" | head; for f in $(find $O/hip -name "*hip_api_trace.csv"); do cp $f $O/hip_api_trace.csv; done; for f in $(find $O/hip -name "*kernel_trace.csv"); do cp $f $O/kernel_trace.csv; done
rm -rf $O/hip; ls -la $O
