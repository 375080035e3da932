#!/bin/bash
R="$GRAFT_REPO_ROOT"; cd "$R" || exit 1
mkdir -p gpurun_out/k
( time timeout 900 python -m pytest tests/test_gpu_workflow.py -m gpu -q --timeout 600 -p no:cacheprovider -k "config1" ) > gpurun_out/k/pytest.log 2>&1
echo "rc=$?"; tail -30 gpurun_out/k/pytest.log | cut -c1-600
