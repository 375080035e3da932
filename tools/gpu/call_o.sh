#!/bin/bash
# e2e probe (plain) + the workflow GPU tests
R="$GRAFT_REPO_ROOT"; cd "$R" || exit 1
O=$R/gpurun_out/o; mkdir -p $O
export PG_E2E_DIR=$R/tools/e2e/_data
export PG_E2E_REPS=8
bash tools/e2e/run.sh 10000 30 ${1:-32} 512 ${2:-8} 1 > $O/base.log 2>&1
tail -1 $O/base.log
python - <<PY
import json, statistics
d=json.load(open("gpurun_out/e2e_probe.json"))
r=d["runs"][2:]
print("median sites/s %.0f" % statistics.median(x["sites_per_s"] for x in r), "cpu %.2f" % statistics.median(x["cpu_user_s"]+x["cpu_sys_s"] for x in r), ["%.0f" % x["sites_per_s"] for x in r])
PY
( time timeout 900 python -m pytest tests/test_gpu_workflow.py tests/test_gpu_parity.py -m gpu -q --timeout 600 -p no:cacheprovider ) > $O/pytest.log 2>&1; tail -5 $O/pytest.log
