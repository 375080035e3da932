#!/bin/bash
# round 3, call K: fresh random inputs on the final tree -- every stage's fuzz tests (incl. the general path and the 32-lane
# wide kernels) under other salts, and the stress parity against the reference's gssw.c
R="$GRAFT_REPO_ROOT"; cd "$R" || exit 1
O=$R/gpurun_out/r03_k; mkdir -p $O
for salt in 711 822 933 1044; do
  PG_SEED_SALT=$salt timeout 900 python -m pytest tests/test_gpu_klib.py tests/test_gpu_parity.py tests/test_gpu_counts.py tests/test_gpu_path.py tests/test_gpu_general.py -m gpu -q -p no:cacheprovider > $O/salt_$salt.log 2>&1
  echo "salt $salt rc=$? $(tail -1 $O/salt_$salt.log)"
done
for seed in 505 606; do
  timeout 900 python tests/stress_parity.py 2000 $seed > $O/stress_$seed.log 2>&1; echo "stress $seed rc=$? $(tail -1 $O/stress_$seed.log)"
done
