#!/bin/bash
# round 6, call z: the exact shortcut's fuzz on 30 more salts
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}; cd "$R"; O=$R/gpurun_out/r6z; mkdir -p $O
for s in $(seq 101 130); do
  PG_SEED_SALT=$s python -m pytest tests/test_gpu_exact.py -m gpu -q -x -p no:cacheprovider -k fuzz 2>&1 | tail -1 | sed "s/^/salt $s: /" | tee -a $O/exact_stress.txt
done
grep -c passed $O/exact_stress.txt
