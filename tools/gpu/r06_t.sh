#!/bin/bash
# round 6, call t: the fuzz comparisons on four new salts (paired trace stores in every variant), then stress_parity on two seeds
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}; cd "$R"; O=$R/gpurun_out/r6t; mkdir -p $O
bash tools/stress_gpu.sh 61 64 2>&1 | tee $O/stress.txt
for seed in 601 602; do timeout 900 python tests/stress_parity.py 3000 $seed 2>&1 | tail -2 | tee -a $O/stress_parity.txt; done
