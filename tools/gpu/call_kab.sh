#!/bin/bash
# A/B timing of klib-stage library variants on ONE box: tools/gpu/call_kab.sh name1 name2 ... (tools/variants/lib_<name>.so)
R="$GRAFT_REPO_ROOT"; cd "$R" || exit 1
mkdir -p gpurun_out/kab; : > gpurun_out/kab/kab.jsonl
for round in 1 2; do for v in "$@"; do
  PG_LIB=$R/tools/variants/lib_$v.so timeout 200 python tools/klib_probe.py 1000000 2>/dev/null | tail -1 | sed "s/^{/{\"variant\": \"$v\", /" | tee -a gpurun_out/kab/kab.jsonl
done; done
