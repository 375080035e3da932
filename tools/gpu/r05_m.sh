#!/bin/bash
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}; cd "$R"; O=$R/gpurun_out/r5m; mkdir -p $O
python tools/e2e/phase_probe.py 6000 path_sequence_matching=1 | tee $O/phase_path.json
python tools/e2e/phase_probe.py 6000 | tee $O/phase_gssw.json
