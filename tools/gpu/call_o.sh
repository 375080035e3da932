#!/bin/bash
# e2e probe: lanes by default (one host thread per lane) + the workflow GPU tests
R="$GRAFT_REPO_ROOT"; cd "$R" || exit 1
O=$R/gpurun_out/o; mkdir -p $O
export PG_E2E_DIR=$R/tools/e2e/_data
W=$PG_E2E_DIR
export PG_E2E_REPS=8
bash tools/e2e/run.sh 10000 30 16 512 0 1 > $O/base.log 2>&1
tail -1 $O/base.log
cp gpurun_out/e2e_probe.json $O/e2e_probe_16.json
run() { name=$1; TH=$2; LN=$3; PB=$4; shift 4; ( env "$@" $W/grmpy_batch $W/ref.fa $W/manifest.txt $W/graphs.txt $TH $W/g_$name.json $PB $LN 1 ) > $O/$name.json 2> $O/$name.err; python - <<PY
import json, statistics
d=json.load(open("$O/$name.json"))
r=d["runs"][2:]
print("$name", "lanes", d["runs"][-1]["lanes"], "batches", d["runs"][-1]["batches"], "median sites/s %.0f" % statistics.median(x["sites_per_s"] for x in r), "cpu %.2f" % statistics.median(x["cpu_user_s"]+x["cpu_sys_s"] for x in r), ["%.0f" % x["sites_per_s"] for x in r])
PY
}
run t16_default 16 0 0 A=1
run t16_default2 16 0 0 A=1
run t24_default 24 0 0 A=1
( time timeout 900 python -m pytest tests/test_gpu_workflow.py -m gpu -q --timeout 600 -p no:cacheprovider ) > $O/pytest.log 2>&1; tail -4 $O/pytest.log
