#!/bin/bash
# round 6: the lean gssw stage's own tests and the parity suite's fuzz tests (which run the lean stage for every chunk) on 30 more salts
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}; cd "$R"; O=$R/gpurun_out/r6leanstress; mkdir -p $O
for s in $(seq 301 330); do
  PG_SEED_SALT=$s timeout 600 python -m pytest tests/test_gpu_lean.py tests/test_gpu_parity.py tests/test_gpu_configs.py -m gpu -q -x -p no:cacheprovider -k "not launch_settings and not two_fill_streams" 2>&1 | tail -1 | sed "s/^/salt $s: /" | tee -a $O/lean_stress.txt
done
grep -c passed $O/lean_stress.txt
