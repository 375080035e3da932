#!/bin/bash
# round 6: the headline step with the plain gssw stage and with the lean one (PG_LEAN=1), then the lean one again with the reference check
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}; cd "$R"; O=$R/gpurun_out/r6lean; mkdir -p $O
run() {
  env "$@" python bench.py --sites-steps 0 --config5-graphs 0 --e2e-steps 0 --exact-shortcut-steps 0 --no-cpu-baseline --stream-batches 0 2> $O/err.txt | python -c "
import json,sys
d=json.loads([l for l in sys.stdin if l.startswith('{')][-1])
print(json.dumps({'env': '$*', 'value': round(d['value']), 'ms_per_step': round(d['ms_per_step'],2), 'kernel_ms': d.get('kernel_ms')}))" | tee -a $O/lean_ab.jsonl
}
run PG_LEAN=0; run PG_LEAN=1; run PG_LEAN=0; run PG_LEAN=1
PG_LEAN=1 python bench.py --sites-steps 0 --config5-graphs 0 --e2e-steps 0 --exact-shortcut-steps 0 --stream-batches 0 2> $O/err2.txt | python -c "
import json,sys
d=json.loads([l for l in sys.stdin if l.startswith('{')][-1])
print(json.dumps({'value': round(d['value']), 'verified': d.get('verified'), 'counts': d.get('counts')})[:1500])" | tee -a $O/lean_verified.json
tail -3 $O/err2.txt
