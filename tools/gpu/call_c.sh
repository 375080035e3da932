#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out/c
timeout 120 tools/ubench/store_bw 50000 > gpurun_out/c/store_bw.txt 2>&1; cat gpurun_out/c/store_bw.txt
for m in 0 1 2 3; do PG_FILL_DEBUG=$m timeout 200 python tools/fill_probe.py 200000 2>/dev/null | tail -1 | tee -a gpurun_out/c/fill_probe.jsonl; done
