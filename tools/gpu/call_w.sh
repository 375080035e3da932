#!/bin/bash
# traceback with a bounded number of wavefronts in flight (PG_TRACE_GRID): what the fill gains
R="$GRAFT_REPO_ROOT"; cd "$R" || exit 1
mkdir -p gpurun_out/w; : > gpurun_out/w/grid.jsonl
for round in 1 2; do for g in 0 8192 4096 2048 1024 512; do
  if [ $g = 0 ]; then unset PG_TRACE_GRID; else export PG_TRACE_GRID=$g; fi
  timeout 300 python tools/overlap_probe.py 1000000 4 0 2>/dev/null | tail -1 | sed "s/^{/{\"grid\": $g, /" | tee -a gpurun_out/w/grid.jsonl | cut -c1-300
done; done
