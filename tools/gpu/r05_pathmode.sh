#!/bin/bash
# `paragraph`'s default cascade in the workflow: e2e leg, per-call host timing and the lanes' timeline of a path-mode pass
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}; cd "$R"; O=$R/gpurun_out/r5pm; mkdir -p $O
(timeout 600 python -m pytest tests -m gpu -x -q -p no:cacheprovider -k "test_gpu_path or test_gpu_workflow") > $O/tests.log 2>&1; echo "tests rc=$? $(grep -E 'passed|failed' $O/tests.log | tail -1)"
for i in 1 2 3; do
python bench.py --reads 20000 --steps 1 --warmup 0 --sites-steps 0 --no-cpu-baseline --stream-batches 0 --e2e-steps 4 --e2e-options '{}' 2> /dev/null | python -c "
import json,sys
d=json.loads([l for l in sys.stdin if l.startswith('{')][-1])['e2e']
print(json.dumps({'sites_genotyped_per_s': round(d['sites_genotyped_per_s']), 'mismatches': d['mismatches'], 'with_path_matching': round(d['with_path_matching']['sites_genotyped_per_s'])}))" | tee -a $O/e2e.jsonl
done
PG_API_TIMING=1 python tools/e2e/mode_trace.py run /dev/shm/pg_t 4 path_sequence_matching=1 > $O/api_timing.json 2> $O/api_timing.err; cat $O/api_timing.json; grep PG_API_TIMING $O/api_timing.err | head -16
python tools/e2e/lane_trace.py path_sequence_matching=1 > $O/lane_trace.json 2> $O/lane_trace.err; python -c "
import json;d=json.load(open('$O/lane_trace.json'));s=d.pop('lanes_in_phase_over_time');print(json.dumps(d));
for x in s:print(x)"
