#!/bin/bash
# how much the kernels on the second stream cost the fill: steady-state probe with library variants, then parity
R="$GRAFT_REPO_ROOT"; cd "$R" || exit 1
mkdir -p gpurun_out/u; : > gpurun_out/u/overlap.jsonl
for round in 1 2; do for v in "$@"; do
  PG_LIB=$R/tools/variants/lib_$v.so timeout 300 python tools/overlap_probe.py 1000000 4 0 2>gpurun_out/u/err_$v.log | tail -1 | sed "s/^{/{\"variant\": \"$v\", /" | tee -a gpurun_out/u/overlap.jsonl
done; done
( time timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_configs.py tests/test_gpu_scale.py -m gpu -q --timeout 800 -p no:cacheprovider -x ) > gpurun_out/u/pytest.log 2>&1; grep -E "passed|failed" gpurun_out/u/pytest.log | tail -2
