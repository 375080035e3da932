#!/bin/bash
# e2e probe: spinning vs blocking host waits
R="$GRAFT_REPO_ROOT"; cd "$R" || exit 1
O=$R/gpurun_out/o; mkdir -p $O
export PG_E2E_DIR=$R/tools/e2e/_data
W=$PG_E2E_DIR
export PG_E2E_REPS=8
bash tools/e2e/run.sh 10000 30 32 512 8 1 > $O/base.log 2>&1
run() { name=$1; shift; ( env "$@" $W/grmpy_batch $W/ref.fa $W/manifest.txt $W/graphs.txt 32 $W/g_$name.json 512 8 1 ) > $O/$name.json 2> $O/$name.err; python - <<PY
import json, statistics
d=json.load(open("$O/$name.json"))
r=d["runs"][2:]
print("$name", "median sites/s %.0f" % statistics.median(x["sites_per_s"] for x in r), "cpu %.2f" % statistics.median(x["cpu_user_s"]+x["cpu_sys_s"] for x in r), ["%.0f" % x["sites_per_s"] for x in r])
PY
}
run spin A=1
run block PG_BLOCKING_SYNC=1
run spin2 A=1
run block2 PG_BLOCKING_SYNC=1
