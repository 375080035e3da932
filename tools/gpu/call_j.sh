#!/bin/bash
R="$GRAFT_REPO_ROOT"; cd "$R" || exit 1
mkdir -p gpurun_out/j
( time timeout 600 python -m pytest tests/test_gpu_parity.py tests/test_gpu_counts.py tests/test_gpu_configs.py -m gpu -x -q --timeout 300 -p no:cacheprovider ) > gpurun_out/j/pytest_fast.log 2>&1
rc=$?; echo "fast rc=$rc"; tail -4 gpurun_out/j/pytest_fast.log
if [ $rc -ne 0 ]; then head -c 8000 gpurun_out/j/pytest_fast.log; exit 1; fi
( time timeout 600 python -m pytest tests/test_gpu_scale.py -m gpu -x -q --timeout 600 -p no:cacheprovider -k "stress" ) > gpurun_out/j/pytest_stress.log 2>&1
echo "stress rc=$?"; tail -4 gpurun_out/j/pytest_stress.log
timeout 300 python tools/readlen_probe.py 2>/dev/null | tail -1 > gpurun_out/j/readlen_probe.json; python -c "
import json; d=json.load(open('gpurun_out/j/readlen_probe.json'))
for r in d['rows']: print(r['read_len'], r['reads_per_s'], r['tcups'], r['device']['fill_ms'], r['device']['trace_ms'])"
