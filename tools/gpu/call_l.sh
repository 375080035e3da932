#!/bin/bash
R="$GRAFT_REPO_ROOT"; cd "$R" || exit 1
mkdir -p gpurun_out/l
( time timeout 900 python -m pytest tests/test_gpu_klib.py -m gpu -q --timeout 600 -p no:cacheprovider ) > gpurun_out/l/pytest_klib.log 2>&1
echo "klib rc=$?"; tail -30 gpurun_out/l/pytest_klib.log | cut -c1-600
( time timeout 1500 python -m pytest tests -m gpu -q --timeout 900 -p no:cacheprovider --deselect tests/test_gpu_klib.py ) > gpurun_out/l/pytest_all.log 2>&1
echo "all rc=$?"; tail -12 gpurun_out/l/pytest_all.log | cut -c1-400
