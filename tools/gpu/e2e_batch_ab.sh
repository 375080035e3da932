#!/bin/bash
# BAM -> genotypes probe at several (site, sample) pairs per device batch and lane counts: tools/gpu/e2e_batch_ab.sh
R="$GRAFT_REPO_ROOT"; cd "$R" || exit 1
O=$R/gpurun_out/e2e_batch_ab; mkdir -p $O; : > $O/ab.jsonl
for cfg in "0 0" "192 0" "256 0" "384 0" "256 32" "0 32"; do
  set -- $cfg
  PG_E2E_DIR=tools/e2e/_data PG_E2E_REPS=8 timeout 300 tools/e2e/run.sh 10000 30 16 $1 $2 1 > $O/e2e_$1_$2.log 2>&1
  PB=$1 LN=$2 python - <<PY | tee -a $O/ab.jsonl
import json, os, statistics as st
d = json.load(open("gpurun_out/e2e_probe.json"))
rs = d["runs"][1:]
print(json.dumps({"sites_per_batch": int(os.environ["PB"]), "lanes_arg": int(os.environ["LN"]), "lanes": rs[0]["lanes"], "batches": rs[0]["batches"], "total_s_median": round(st.median(r["total_s"] for r in rs), 4), "sites_per_s_median": round(st.median(r["sites_per_s"] for r in rs)), "cpu_s_median": round(st.median(r["cpu_user_s"] + r["cpu_sys_s"] for r in rs), 3), "device_lane_s": round(st.median(r["device_batch_s"] for r in rs), 3)}))
PY
done
