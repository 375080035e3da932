#!/bin/bash
# steady-state probe: with and without the count kernels on the second stream
R="$GRAFT_REPO_ROOT"; cd "$R" || exit 1
mkdir -p gpurun_out/u; : > gpurun_out/u/overlap.jsonl
for round in 1 2; do for cnt in 1 0; do
  timeout 300 python tools/overlap_probe.py 1000000 4 $cnt 2>gpurun_out/u/err.log | tail -1 | tee -a gpurun_out/u/overlap.jsonl | cut -c1-260
done; done
tail -2 gpurun_out/u/err.log
