#!/bin/bash
R="$GRAFT_REPO_ROOT"; cd "$R" || exit 1
bash tools/sq_collect.sh > gpurun_out/sq_collect.log 2>&1
for s in set1 set2 set3; do python - <<PY
import csv,collections
rows=list(csv.DictReader(open('gpurun_out/sq/$s/sq_counter_collection.csv')))
agg=collections.defaultdict(lambda: collections.defaultdict(float))
for r in rows:
    if 'pg_fill' in r['Kernel_Name']:
        agg[r['Kernel_Name'][:40]][r['Counter_Name']]+=float(r['Counter_Value'])
for k,v in agg.items(): print(k, dict(v))
PY
done
