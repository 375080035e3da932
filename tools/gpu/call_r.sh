#!/bin/bash
R="$GRAFT_REPO_ROOT"; cd "$R" || exit 1
O=$R/gpurun_out/r; mkdir -p $O
( time timeout 600 python -m pytest tests/test_gpu_klib.py tests/test_gpu_workflow.py tests/test_gpu_parity.py -m gpu -q --timeout 500 -p no:cacheprovider ) > $O/pytest.log 2>&1; grep -E "passed|failed|Error|error" $O/pytest.log | tail -5
PG_SEED_SALT=123 timeout 600 python -m pytest tests/test_gpu_klib.py -m gpu -q --timeout 500 -p no:cacheprovider 2>&1 | tail -1
PG_SEED_SALT=456 timeout 600 python -m pytest tests/test_gpu_klib.py -m gpu -q --timeout 500 -p no:cacheprovider 2>&1 | tail -1
