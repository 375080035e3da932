#!/bin/bash
# round 5, call g: the workflow's phases with and without path matching; the pipelined stage probe after the list kernel's wave-aggregated atomics; the cascade tests
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}; cd "$R"; O=$R/gpurun_out/r5g; mkdir -p $O
(time timeout 900 python -m pytest tests -m gpu -x -q -p no:cacheprovider -k "cascade or hand_over or test_gpu_workflow or host_cpp") > $O/tests.log 2>&1; echo "tests rc=$? $(grep -E 'passed|failed' $O/tests.log | tail -1)"
python tools/stage_probe.py 1000000 > $O/stage_probe.json 2> $O/stage_probe.err; echo "probe rc=$?"; python -c "
import json; d=json.load(open('$O/stage_probe.json')); print({k: (round(v['reads_per_s']/1e6,2), round(v.get('vs_predicted',0),3)) for k,v in d.items() if isinstance(v,dict)})"
python tools/e2e/phase_probe.py 6000 | tee $O/phase_gssw.json
python tools/e2e/phase_probe.py 6000 path_sequence_matching=1 | tee $O/phase_path.json
