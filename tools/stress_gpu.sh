#!/bin/bash
# Re-run the GPU fuzz comparisons (every stage against its CPU checker) on fresh random inputs.
#   ./tools/stress_gpu.sh [first_salt] [last_salt]
A=${1:-1}; B=${2:-10}
for s in $(seq $A $B); do
    PG_SEED_SALT=$s python -m pytest tests -m gpu -q -x -k "fuzz or tiny_nodes or word_mode or boundaries or 150bp or long_reads" 2>&1 | tail -1 | sed "s/^/salt $s: /"
done
