#!/bin/bash
# round 6: the lean stage with three workspace regions (PG_FILL_STREAMS=2) and a bounded forward grid: is the period of a chunk the
# chain rev -> forward -> traceback -> second look through its region, or the machine?
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}; cd "$R"; O=$R/gpurun_out/r6leanreg; mkdir -p $O
run() {
  env "$@" python bench.py --sites-steps 0 --config5-graphs 0 --e2e-steps 0 --exact-shortcut-steps 0 --no-cpu-baseline --stream-batches 0 --plain-steps 0 2> $O/err.txt | python -c "
import json,sys
d=json.loads([l for l in sys.stdin if l.startswith('{')][-1])
print(json.dumps({'env': '$*', 'value': round(d['value']), 'ms_per_step': round(d['ms_per_step'],2), 'kernel_ms': d.get('kernel_ms')}))" | tee -a $O/lean_regions.jsonl
}
run A=1
run PG_FILL_STREAMS=2
run PG_FILL_STREAMS=2
run PG_LEAN_ONE_STREAM=1
run PG_LEAN=0
run A=1
python bench.py --sites-steps 0 --config5-graphs 0 --e2e-steps 0 --exact-shortcut-steps 0 --stream-batches 0 2> $O/err2.txt | python -c "
import json,sys
d=json.loads([l for l in sys.stdin if l.startswith('{')][-1])
print(json.dumps({'value': round(d['value']), 'verified': d.get('verified'), 'plain': {k: d['plain_stage'][k] for k in ('reads_per_s','records_differing_from_the_lean_step','cigar_strings_equal')}})[:900])"
