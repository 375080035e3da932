#!/bin/bash
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}; cd "$R"; O=$R/gpurun_out/r6j; mkdir -p $O
for rep in 1 2; do for v in "" tools/variants/lib_pathw6.so tools/variants/lib_pathw8.so; do
  env ${v:+PG_LIB=$v} python tools/path_probe.py 1000000 | tee -a $O/path_occupancy.jsonl
done; done
