#!/bin/bash
# whole GPU suite + default bench
R="$GRAFT_REPO_ROOT"; cd "$R" || exit 1
O=$R/gpurun_out/p; mkdir -p $O
export PG_BENCH_VERBOSE=1
( time timeout 1500 python -m pytest tests -m gpu -q --timeout 900 -p no:cacheprovider ) > $O/pytest.log 2>&1
echo "pytest rc=$?"; tail -6 $O/pytest.log
( time timeout 900 python bench.py ) > $O/bench_default.json 2> $O/bench_default.err
echo "bench rc=$?"; tail -c 1500 $O/bench_default.json; echo
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -2
