#!/bin/bash
# klib stage at several read lengths (rows-per-lane classes C = 8, 10, 12, 16)
R="$GRAFT_REPO_ROOT"; cd "$R" || exit 1
O=$R/gpurun_out/r; mkdir -p $O; : > $O/klib_readlen.jsonl
for L in 100 150 180 250; do
  ${PGENV:-env} timeout 300 python tools/klib_probe.py 400000 $L 2>/dev/null | tail -1 | tee -a $O/klib_readlen.jsonl
done
