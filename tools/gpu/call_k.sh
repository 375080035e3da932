#!/bin/bash
R="$GRAFT_REPO_ROOT"; cd "$R" || exit 1
mkdir -p gpurun_out/k
( time timeout 900 python -m pytest tests/test_gpu_workflow.py tests/test_gpu_parity.py -m gpu -q --timeout 600 -p no:cacheprovider ) > gpurun_out/k/pytest.log 2>&1
echo "rc=$?"; tail -25 gpurun_out/k/pytest.log
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -2
