#!/bin/bash
# round-2 final measurements: whole GPU suite, default bench, 2-rank benches, then the profile call
R="$GRAFT_REPO_ROOT"; cd "$R" || exit 1
O=$R/gpurun_out/final; mkdir -p $O
export PG_BENCH_VERBOSE=1
( time timeout 1500 python -m pytest tests -m gpu -q --timeout 900 -p no:cacheprovider ) > $O/pytest.log 2>&1
echo "pytest rc=$?"; tail -6 $O/pytest.log
( time timeout 900 python bench.py ) > $O/bench_default.json 2> $O/bench_default.err
echo "bench rc=$?"; tail -c 600 $O/bench_default.json; echo
( time timeout 900 python bench.py --gpus 2 --workload config3 --steps 3 --warmup 1 --workspace-gib 48 ) > $O/bench_2rank_config3.json 2> $O/bench_2rank_config3.err
echo "bench2 rc=$?"; tail -c 900 $O/bench_2rank_config3.json; echo
timeout 300 python tools/readlen_probe.py 2>/dev/null | tail -1 > $O/readlen_probe.json
timeout 300 python tools/config5_probe.py 2>/dev/null | tail -1 > $O/config5_probe.json; cat $O/config5_probe.json
timeout 300 python tools/stage_probe.py 2>/dev/null | tail -1 > $O/stage_probe.json; head -c 1500 $O/stage_probe.json; echo
bash tools/gpu/call_f.sh
