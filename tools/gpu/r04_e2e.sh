#!/bin/bash
# round 4: the BAM -> genotypes probe on the data set shipped under tools/e2e/_data (python tools/e2e/make_sites.py tools/e2e/_data 10000 30 1),
# once plain (8 passes -> e2e_probe.json) and once under the SIGPROF sampler (30 passes -> prof_report.txt).  tools/gpu/r04_e2e.sh [tag]
R="$GRAFT_REPO_ROOT"; cd "$R" || exit 1
O=$R/gpurun_out/r04_e2e${1:+_$1}; mkdir -p $O
PG_E2E_DIR=tools/e2e/_data PG_E2E_REPS=8 timeout 600 tools/e2e/run.sh 10000 30 16 0 0 1 > $O/e2e.log 2>&1
echo "rc=$?"; tail -2 $O/e2e.log | cut -c1-300
cp gpurun_out/e2e_probe.json $O/e2e_probe.json
python - <<PY
import json
d = json.load(open("$O/e2e_probe.json"))
for r in d["runs"][1:]:
    print("total %.4f s  cpu %.3f+%.3f s  extract %.3f load %.3f device %.3f docs %.3f gt %.3f  sites/s %.0f" % (r["total_s"], r["cpu_user_s"], r["cpu_sys_s"], r["extract_reads_s"], r["load_graphs_s"], r["device_batch_s"], r["documents_s"], r["genotypes_s"], r["sites_per_s"]))
PY
PG_E2E_PROF=$O/prof.txt PG_E2E_DIR=tools/e2e/_data PG_E2E_REPS=30 timeout 600 tools/e2e/run.sh 10000 30 16 0 0 1 > $O/e2e_prof.log 2>&1
python tools/e2e/prof_report.py $O/prof.txt 70 > $O/prof_report.txt 2>&1
gzip -f $O/prof.txt
sed -n '/^leaf/,$p' $O/prof_report.txt | head -80
