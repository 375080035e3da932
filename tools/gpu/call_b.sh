#!/bin/bash
# staged GPU call: fast parity first (stop on failure), then the rest of the suite, then the benches
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out/b
export PG_BENCH_VERBOSE=1
( time timeout 600 python -m pytest tests/test_gpu_parity.py tests/test_gpu_counts.py tests/test_gpu_configs.py -m gpu -x -q --timeout 300 -p no:cacheprovider ) > gpurun_out/b/pytest_fast.log 2>&1
rc=$?; echo "fast rc=$rc"; tail -8 gpurun_out/b/pytest_fast.log
if [ $rc -ne 0 ]; then head -c 6000 gpurun_out/b/pytest_fast.log; exit 1; fi
( time timeout 1500 python -m pytest tests -m gpu -q --timeout 900 -p no:cacheprovider --deselect tests/test_gpu_parity.py --deselect tests/test_gpu_counts.py --deselect tests/test_gpu_configs.py ) > gpurun_out/b/pytest_rest.log 2>&1
echo "rest rc=$?"; tail -12 gpurun_out/b/pytest_rest.log
( time timeout 900 python bench.py ) > gpurun_out/b/bench_default.json 2> gpurun_out/b/bench_default.err
echo "bench rc=$?"; tail -c 4000 gpurun_out/b/bench_default.json; tail -5 gpurun_out/b/bench_default.err
( time timeout 900 python bench.py --gpus 2 --steps 3 --warmup 1 --workspace-gib 48 --sites 4000 ) > gpurun_out/b/bench_2rank.json 2> gpurun_out/b/bench_2rank.err
echo "bench2 rc=$?"; tail -c 2500 gpurun_out/b/bench_2rank.json; tail -5 gpurun_out/b/bench_2rank.err
