#!/bin/bash
# four-sample manifest over the 10 000-site data set (40 000 (site, sample) pairs)
R="$GRAFT_REPO_ROOT"; cd "$R" || exit 1
O=$R/gpurun_out/o; mkdir -p $O
export PG_E2E_DIR=$R/tools/e2e/_data
W=$PG_E2E_DIR
export PG_E2E_REPS=4
bash tools/e2e/run.sh 10000 30 16 0 0 1 > $O/base.log 2>&1; tail -1 $O/base.log
printf 'id\tpath\tdepth\tread length\n' > $O/manifest4.txt
for s in A B C D; do printf 'S%s\ttools/e2e/_data/reads.bam\t30\t150\n' $s >> $O/manifest4.txt; done
( time $W/grmpy_batch $W/ref.fa $O/manifest4.txt $W/graphs.txt 16 /tmp/g4.json 0 0 1 ) > $O/four.json 2> $O/four.err; python - <<PY
import json
d=json.load(open("$O/four.json")); r=d["runs"]
print("4 samples:", [("%.3f s" % x["total_s"], "%.0f pairs/s" % (4*x["sites"]/x["total_s"]), "%.1f M reads/s" % (x["reads_per_s"]/1e6)) for x in r], "lanes", r[-1]["lanes"], "batches", r[-1]["batches"])
PY
tail -2 $O/four.err
