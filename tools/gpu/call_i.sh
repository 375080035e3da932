#!/bin/bash
R="$GRAFT_REPO_ROOT"; cd "$R" || exit 1
mkdir -p gpurun_out/i
( time timeout 600 python -m pytest tests/test_gpu_parity.py tests/test_gpu_counts.py tests/test_gpu_configs.py -m gpu -x -q --timeout 300 -p no:cacheprovider ) > gpurun_out/i/pytest_fast.log 2>&1
rc=$?; echo "fast rc=$rc"; tail -4 gpurun_out/i/pytest_fast.log
if [ $rc -ne 0 ]; then head -c 6000 gpurun_out/i/pytest_fast.log; exit 1; fi
bash tools/gpu/call_ab.sh "$@"
( time timeout 600 python -m pytest tests/test_gpu_scale.py -m gpu -x -q --timeout 600 -p no:cacheprovider -k "not bench" ) > gpurun_out/i/pytest_scale.log 2>&1
echo "scale rc=$?"; tail -4 gpurun_out/i/pytest_scale.log
