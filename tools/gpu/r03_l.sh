#!/bin/bash
# round 3, call L: sampled CPU profile of the BAM -> genotypes probe (tools/e2e/prof.hh), to see what the host side spends now
R="$GRAFT_REPO_ROOT"; cd "$R" || exit 1
O=$R/gpurun_out/r03_l; mkdir -p $O
PG_E2E_PROF=$O/prof.txt PG_E2E_DIR=tools/e2e/_data PG_E2E_REPS=8 timeout 600 tools/e2e/run.sh 10000 30 16 0 0 1 > $O/e2e.log 2>&1
echo "rc=$?"; tail -2 $O/e2e.log | cut -c1-200
python tools/e2e/prof_report.py $O/prof.txt 80 > $O/prof_report.txt 2>&1
head -100 $O/prof_report.txt
gzip -f $O/prof.txt
