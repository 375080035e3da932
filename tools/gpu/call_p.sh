#!/bin/bash
# whole GPU suite + smoke
R="$GRAFT_REPO_ROOT"; cd "$R" || exit 1
O=$R/gpurun_out/p; mkdir -p $O
( time timeout 1500 python -m pytest tests -m gpu -q --timeout 900 -p no:cacheprovider ) > $O/pytest.log 2>&1
echo "pytest rc=$?"; grep -E "passed|failed" $O/pytest.log | tail -2
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -2
