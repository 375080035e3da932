#!/bin/bash
R="$GRAFT_REPO_ROOT"; cd "$R" || exit 1
mkdir -p gpurun_out/l
( time timeout 900 python -m pytest tests/test_gpu_klib.py -m gpu -q --timeout 600 -p no:cacheprovider ) > gpurun_out/l/pytest_klib.log 2>&1
echo "klib rc=$?"; tail -30 gpurun_out/l/pytest_klib.log | cut -c1-600
bash tools/gpu/call_m.sh
