#!/bin/bash
# fresh random inputs: the klib tests and the stress parity under other salts
R="$GRAFT_REPO_ROOT"; cd "$R" || exit 1
O=$R/gpurun_out/t; mkdir -p $O
for salt in 101 202 303 404 505 606; do
  PG_SEED_SALT=$salt timeout 600 python -m pytest tests/test_gpu_klib.py tests/test_gpu_parity.py tests/test_gpu_counts.py tests/test_gpu_path.py -m gpu -q --timeout 500 -p no:cacheprovider -x > $O/salt_$salt.log 2>&1
  echo "salt $salt rc=$? $(tail -1 $O/salt_$salt.log)"
done
for seed in 303 404; do
  timeout 900 python tests/stress_parity.py 2000 $seed > $O/stress_$seed.log 2>&1; echo "stress $seed rc=$? $(tail -1 $O/stress_$seed.log)"
done
