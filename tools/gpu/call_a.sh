#!/bin/bash
# round-2 GPU call A: parity suite (incl. full-size tests), default bench, 2-rank bench on the shared GPU
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out/a
export PG_BENCH_VERBOSE=1
( time timeout 1500 python -m pytest tests -m gpu -q --timeout 900 -p no:cacheprovider ) > gpurun_out/a/pytest.log 2>&1
echo "pytest rc=$?" >> gpurun_out/a/pytest.log
tail -15 gpurun_out/a/pytest.log
( time timeout 900 python bench.py ) > gpurun_out/a/bench_default.json 2> gpurun_out/a/bench_default.err
echo "bench rc=$?"; tail -c 3000 gpurun_out/a/bench_default.json; tail -5 gpurun_out/a/bench_default.err
( time timeout 900 python bench.py --gpus 2 --steps 3 --warmup 1 --workspace-gib 48 ) > gpurun_out/a/bench_2rank.json 2> gpurun_out/a/bench_2rank.err
echo "bench2 rc=$?"; tail -c 1500 gpurun_out/a/bench_2rank.json; tail -5 gpurun_out/a/bench_2rank.err
