#!/bin/bash
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}; cd "$R"; O=$R/gpurun_out/r5kr; mkdir -p $O
(time timeout 1200 python -m pytest tests -m gpu -x -q -p no:cacheprovider -k "test_gpu_kmer or test_gpu_klib or test_gpu_counts or test_gpu_path") > $O/tests.log 2>&1; echo "tests rc=$? $(grep -E 'passed|failed' $O/tests.log | tail -1)"; grep -E "Error|assert" $O/tests.log | head -5
timeout 600 python tools/stage_probe.py 1000000 > $O/stage_probe.json 2> $O/stage_probe.err; echo "probe rc=$?"; python -c "
import json;d=json.load(open('$O/stage_probe.json'));print({k:(round(v['reads_per_s']/1e6,2) if isinstance(v,dict) and 'reads_per_s' in v else None) for k,v in d.items()})"
