#!/bin/bash
R="$GRAFT_REPO_ROOT"; cd "$R" || exit 1
mkdir -p gpurun_out/h
( time timeout 600 python -m pytest tests/test_gpu_parity.py tests/test_gpu_counts.py tests/test_gpu_configs.py -m gpu -x -q --timeout 300 -p no:cacheprovider ) > gpurun_out/h/pytest_fast.log 2>&1
rc=$?; echo "fast rc=$rc"; tail -4 gpurun_out/h/pytest_fast.log
if [ $rc -ne 0 ]; then head -c 6000 gpurun_out/h/pytest_fast.log; exit 1; fi
timeout 200 python tools/fill_probe.py 200000 2>/dev/null | tail -1 | tee gpurun_out/h/fill_probe.json
bash tools/gpu/call_g.sh
