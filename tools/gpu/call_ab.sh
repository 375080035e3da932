#!/bin/bash
# A/B timing of library variants on ONE box: tools/gpu/call_ab.sh name1 name2 ... (tools/variants/lib_<name>.so), 3 interleaved rounds
R="$GRAFT_REPO_ROOT"; cd "$R" || exit 1
mkdir -p gpurun_out/ab; : > gpurun_out/ab/ab.jsonl
for round in 1 2 3; do for v in "$@"; do
  PG_LIB=$R/tools/variants/lib_$v.so timeout 200 python tools/fill_probe.py 200000 2>/dev/null | tail -1 | sed "s/^{/{\"variant\": \"$v\", /" | tee -a gpurun_out/ab/ab.jsonl
done; done
