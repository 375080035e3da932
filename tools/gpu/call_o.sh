#!/bin/bash
# e2e probe: cold first pass and steady state + workflow tests
R="$GRAFT_REPO_ROOT"; cd "$R" || exit 1
O=$R/gpurun_out/o; mkdir -p $O
export PG_E2E_DIR=$R/tools/e2e/_data
W=$PG_E2E_DIR
export PG_E2E_REPS=5
bash tools/e2e/run.sh 10000 30 16 0 0 1 > $O/base.log 2>&1; tail -1 $O/base.log
for i in 1 2 3; do ( time $W/grmpy_batch $W/ref.fa $W/manifest.txt $W/graphs.txt 16 $W/g_x.json 0 0 1 ) > $O/x$i.json 2> $O/x$i.err; python - <<PY
import json
d=json.load(open("$O/x$i.json")); r=d["runs"]
print("process $i: first pass %.3f s, then" % r[0]["total_s"], ["%.3f" % x["total_s"] for x in r[1:]], "pinned MB", r[-1]["pinned_staging_bytes"]>>20)
PY
grep real $O/x$i.err; done
( time timeout 900 python -m pytest tests/test_gpu_workflow.py -m gpu -q --timeout 600 -p no:cacheprovider ) > $O/pytest.log 2>&1; grep -E "passed|failed" $O/pytest.log
