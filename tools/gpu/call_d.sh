#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out/d
export PG_BENCH_VERBOSE=1
( time timeout 600 python -m pytest tests/test_gpu_parity.py tests/test_gpu_counts.py tests/test_gpu_configs.py tests/test_gpu_workflow.py -m gpu -x -q --timeout 300 -p no:cacheprovider ) > gpurun_out/d/pytest_fast.log 2>&1
rc=$?; echo "fast rc=$rc"; tail -8 gpurun_out/d/pytest_fast.log
if [ $rc -ne 0 ]; then head -c 6000 gpurun_out/d/pytest_fast.log; exit 1; fi
timeout 200 python tools/fill_probe.py 200000 2>/dev/null | tail -1 | tee gpurun_out/d/fill_probe.json
( time timeout 600 python -m pytest tests/test_gpu_scale.py -m gpu -x -q --timeout 600 -p no:cacheprovider ) > gpurun_out/d/pytest_scale.log 2>&1
echo "scale rc=$?"; tail -6 gpurun_out/d/pytest_scale.log
( time timeout 900 python bench.py --sites-steps 0 ) > gpurun_out/d/bench_default.json 2> gpurun_out/d/bench_default.err
echo "bench rc=$?"; tail -c 1800 gpurun_out/d/bench_default.json; tail -3 gpurun_out/d/bench_default.err
