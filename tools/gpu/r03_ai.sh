#!/bin/bash
# round 3, call AI: runtime settings against the host CPU of the BAM -> genotypes probe
R="$GRAFT_REPO_ROOT"; cd "$R" || exit 1
O=$R/gpurun_out/r03_ai; mkdir -p $O
PG_E2E_DIR=tools/e2e/_data PG_E2E_REPS=2 timeout 600 tools/e2e/run.sh 10000 30 16 0 0 1 > $O/build.log 2>&1
W=tools/e2e/_data
run() {
  local name=$1; shift
  env "$@" PG_E2E_REPS=6 timeout 300 $W/grmpy_batch $W/ref.fa $W/manifest.txt $W/graphs.txt 16 $O/g_$name.json 0 0 1 > $O/$name.json 2> $O/$name.err
  python - $O/$name.json $name <<'PY'
import json, sys
try:
    runs = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])["runs"][1:]
    cpu = sorted(r["cpu_user_s"] + r["cpu_sys_s"] for r in runs)[len(runs) // 2]
    tot = sorted(r["total_s"] for r in runs)[len(runs) // 2]
    print(sys.argv[2], "median total_s %.3f cpu %.2f" % (tot, cpu))
except Exception as e:
    print(sys.argv[2], "failed", e)
PY
}
run default A=1
run nointerrupt HSA_ENABLE_INTERRUPT=0
run activewait ROC_ACTIVE_WAIT_TIMEOUT=50
run nodirect AMD_DIRECT_DISPATCH=0
run spin PG_SPIN_WAITS=1
run default2 A=1
