#!/bin/bash
# round 6, call y: the exact shortcut in front of the gssw stage
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}; cd "$R"; O=$R/gpurun_out/r6y; mkdir -p $O
(time timeout 1200 python -m pytest tests -m gpu -x -q -p no:cacheprovider -k "test_gpu_exact or test_gpu_path") > $O/tests.log 2>&1; echo "tests rc=$? $(grep -E 'passed|failed' $O/tests.log | tail -1)"; grep -E "^E" $O/tests.log | cut -c1-600 | head -20
