#!/bin/bash
# round 6, call r: -A / filtered alignments test + workflow tests; the 12-row variant (161 - 192 bp reads) with paired trace stores against the tree before
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}; cd "$R"; O=$R/gpurun_out/r6r; mkdir -p $O
(time timeout 1200 python -m pytest tests -m gpu -x -q -p no:cacheprovider -k "test_gpu_workflow or host_cpp") > $O/tests.log 2>&1; echo "tests rc=$? $(grep -E 'passed|failed' $O/tests.log | tail -1)"; grep -E "^E|Error" $O/tests.log | head -20
for rep in 1 2; do for v in tools/variants/lib_base6.so ""; do
  env ${v:+PG_LIB=$v} python tools/readlen_probe.py 200000 150,180,250 | sed "s|^{|{\"lib\": \"${v:-tree}\", |" | tee -a $O/readlen_ab.jsonl
done; done
