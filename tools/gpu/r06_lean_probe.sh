#!/bin/bash
# round 6, timing probes of the lean stage: what the second look costs (skipped: records of the listed reads missing), the traceback's grid
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}; cd "$R"; O=$R/gpurun_out/r6leanprobe; mkdir -p $O
run() {
  env "$@" python bench.py --sites-steps 0 --config5-graphs 0 --e2e-steps 0 --exact-shortcut-steps 0 --no-cpu-baseline --stream-batches 0 --plain-steps 0 2> $O/err.txt | python -c "
import json,sys
d=json.loads([l for l in sys.stdin if l.startswith('{')][-1])
print(json.dumps({'env': '$*', 'value': round(d['value']), 'ms_per_step': round(d['ms_per_step'],2), 'kernel_ms': d.get('kernel_ms')}))" | tee -a $O/lean_probe.jsonl
}
run A=1
run PG_LEAN_TIMING_SKIP_SECOND=1
run PG_TRACE_BLOCKS=2048
run PG_TRACE_BLOCKS=6144
run PG_LEAN_TIMING_SKIP_SECOND=1 PG_FILL_STREAMS=2 PG_LEAN_INST_BLOCKS=6
run A=1
