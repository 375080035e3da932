#!/bin/bash
# round 3, call AE: the BAM -> genotypes probe on the final kernels, and the KFD calls it makes (LD_PRELOAD counter)
R="$GRAFT_REPO_ROOT"; cd "$R" || exit 1
O=$R/gpurun_out/r03_ae; mkdir -p $O
gcc -O2 -shared -fPIC -o $O/ioctl_count.so tools/e2e/ioctl_count.c -ldl
PG_E2E_DIR=tools/e2e/_data PG_E2E_REPS=6 timeout 600 tools/e2e/run.sh 10000 30 16 0 0 1 > $O/e2e.log 2>&1
echo "rc=$?"; tail -1 $O/e2e.log | cut -c1-200
python - $O/e2e.log <<'PY'
import json, sys
for l in open(sys.argv[1]):
    if l.startswith('{"graphs"'):
        runs = json.loads(l)["runs"][1:]
        for r in runs:
            print("total_s %.3f sites/s %.0f cpu %.2f (sys %.2f)" % (r["total_s"], r["sites_per_s"], r["cpu_user_s"] + r["cpu_sys_s"], r["cpu_sys_s"]))
PY
W=tools/e2e/_data
LD_PRELOAD=$O/ioctl_count.so PG_E2E_REPS=6 $W/grmpy_batch $W/ref.fa $W/manifest.txt $W/graphs.txt 16 $O/genotypes.json 0 0 1 > $O/ioctl_run.json 2> $O/ioctl.txt
grep "ioctl nr 0x0c" $O/ioctl.txt
