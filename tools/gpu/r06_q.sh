#!/bin/bash
# round 6, call q: trace stores as pairs of steps (global_store_dwordx2): fill / traceback alone A/B against the tree before (lib_base6), then the whole GPU suite
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}; cd "$R"; O=$R/gpurun_out/r6q; mkdir -p $O
for rep in 1 2 3; do for v in tools/variants/lib_base6.so ""; do
  env ${v:+PG_LIB=$v} python tools/fill_probe.py 200000 | sed "s|^{|{\"lib\": \"${v:-tree}\", |" | tee -a $O/trace_pairs_ab.jsonl
done; done
(time timeout 1800 python -m pytest tests -m gpu -x -q -p no:cacheprovider) > $O/tests.log 2>&1; echo "tests rc=$? $(grep -E 'passed|failed' $O/tests.log | tail -1)"
