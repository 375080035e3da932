#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out/e
( time timeout 600 python -m pytest tests/test_gpu_parity.py tests/test_gpu_counts.py tests/test_gpu_configs.py tests/test_gpu_workflow.py -m gpu -x -q --timeout 300 -p no:cacheprovider ) > gpurun_out/e/pytest_fast.log 2>&1
rc=$?; echo "fast rc=$rc"; tail -8 gpurun_out/e/pytest_fast.log
if [ $rc -ne 0 ]; then head -c 6000 gpurun_out/e/pytest_fast.log; exit 1; fi
( time timeout 600 python -m pytest tests/test_gpu_scale.py -m gpu -x -q --timeout 600 -p no:cacheprovider -k "not bench" ) > gpurun_out/e/pytest_scale.log 2>&1
echo "scale rc=$?"; tail -6 gpurun_out/e/pytest_scale.log
timeout 200 python tools/fill_probe.py 200000 2>/dev/null | tail -1 | tee gpurun_out/e/fill_probe.json
timeout 300 python tools/readlen_probe.py 2>/dev/null | tail -1 > gpurun_out/e/readlen_probe.json; head -c 3000 gpurun_out/e/readlen_probe.json; echo
timeout 300 python tools/config5_probe.py 2>/dev/null | tail -1 > gpurun_out/e/config5_probe.json; head -c 1500 gpurun_out/e/config5_probe.json; echo
