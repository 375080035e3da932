#!/bin/bash
# round 3, call AN (information only, after the counters of the final sources were collected): the walk's H loads as non-temporal
# loads (tools/variants/lib_ntwalk.so = the tree with __builtin_nontemporal_load in Hcell, built with tools/build_variant.sh) against the tree
R="$GRAFT_REPO_ROOT"; cd "$R" || exit 1
for V in tree ntwalk tree ntwalk; do
  if [ $V = tree ]; then unset PG_LIB; else export PG_LIB=$R/tools/variants/lib_$V.so; fi
  python bench.py --steps 8 --warmup 2 --no-cpu-baseline --stream-batches 0 --sites-steps 0 --collective off 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$V', round(d['value']/1e6,3), round(d['roofline']['avg_launch_ms'],3))"
done
