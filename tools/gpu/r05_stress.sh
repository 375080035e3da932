#!/bin/bash
# round 5: the GPU fuzz comparisons on fresh random inputs (five salts), the cascade hand-over and the two-fill-stream tests with them
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}; cd "$R"; O=$R/gpurun_out/r5s; mkdir -p $O
for s in 7101 7202 7303 7404 7505; do
  PG_SEED_SALT=$s timeout 600 python -m pytest tests -m gpu -q -x -p no:cacheprovider -k "fuzz or tiny_nodes or word_mode or boundaries or 150bp or long_reads or hand_over or cascade or two_fill_streams" 2>&1 | grep -E "passed|failed" | tail -1 | sed "s/^/salt $s: /" | tee -a $O/stress.txt
done
python tests/stress_parity.py 3000 1717 2>&1 | tail -1 | tee -a $O/stress.txt
