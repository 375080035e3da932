#!/bin/bash
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}; cd "$R"; O=$R/gpurun_out/r6d; mkdir -p $O
python tools/boundary_probe.py 200000 2>$O/err.txt | tee $O/boundary_probe.json
