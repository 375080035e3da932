#!/bin/bash
# round 4, call C: where the fill's time goes now (timing-only builds, tools/experiments/): without the H-trace stores, without
# stores + packing, with the stores only in lanes whose node maximum grew, without stores + first-column tracking
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/r04_c
mkdir -p "$O"
cd "$R"
for round in 1 2; do for v in iadd nostore nostore_noperm maskedstore nostore_nofc; do
  PG_LIB=$R/tools/variants/lib_$v.so timeout 200 python tools/fill_probe.py 200000 2>/dev/null | tail -1 | sed "s/^{/{\"variant\": \"$v\", /" | tee -a "$O/ab.jsonl"
done; done
