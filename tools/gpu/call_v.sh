#!/bin/bash
# PMC traffic of the working tree (fill + traceback HBM bytes per read)
R="$GRAFT_REPO_ROOT"; cd "$R" || exit 1
O=$R/gpurun_out/v; mkdir -p $O
bash tools/pmc_collect.sh r02 > $O/pmc_collect.log 2>&1; python tools/pmc_traffic.py gpurun_out/pmc_r02 $O/traffic.json 200000 > $O/pmc_traffic.log 2>&1; grep -A6 "pg_trace_kernel\|pg_fill_kernel" $O/traffic.json | head -30
