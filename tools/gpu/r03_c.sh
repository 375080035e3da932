#!/bin/bash
# round 3, call C: general path + per-site isolation
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/r03_c
mkdir -p "$O"
cd "$R"
timeout 1200 python -m pytest tests/test_gpu_general.py tests/test_gpu_parity.py "tests/test_gpu_workflow.py::test_one_oversize_site_does_not_take_the_run_down" -m gpu -q > "$O/pytest.log" 2>&1
echo "pytest rc=$?"; tail -40 "$O/pytest.log"
