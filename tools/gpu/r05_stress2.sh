#!/bin/bash
# the rewritten path and k-mer kernels (and everything that runs through them) on fresh random inputs: four salts
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}; cd "$R"; O=$R/gpurun_out/r5s3; mkdir -p $O
for s in 8111 8222 8333 8444; do
  PG_SEED_SALT=$s timeout 600 python -m pytest tests -m gpu -q -x -p no:cacheprovider -k "test_gpu_path or test_gpu_kmer or test_gpu_klib or fuzz or hand_over or cascade or reused" 2>&1 | grep -E "passed|failed|Error" | tail -2 | sed "s/^/salt $s: /" | tee -a $O/stress.txt
done
