#!/bin/bash
# round 6: the lean stage's forward launch beside the next chunk's reversed-graph fills: priority of its stream, wavefronts it holds per CU
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}; cd "$R"; O=$R/gpurun_out/r6leanmix; mkdir -p $O
run() {
  env "$@" python bench.py --sites-steps 0 --config5-graphs 0 --e2e-steps 0 --exact-shortcut-steps 0 --no-cpu-baseline --stream-batches 0 2> $O/err.txt | python -c "
import json,sys
d=json.loads([l for l in sys.stdin if l.startswith('{')][-1])
print(json.dumps({'env': '$*', 'value': round(d['value']), 'ms_per_step': round(d['ms_per_step'],2), 'kernel_ms': d.get('kernel_ms')}))" | tee -a $O/lean_mix.jsonl
}
run PG_LEAN=1
run PG_LEAN=1 PG_FILL2_HIGH=1
run PG_LEAN=1 PG_FILL2_HIGH=1 PG_LEAN_INST_BLOCKS=4
run PG_LEAN=1 PG_FILL2_HIGH=1 PG_LEAN_INST_BLOCKS=6
run PG_LEAN=1 PG_FILL2_HIGH=1 PG_LEAN_INST_BLOCKS=8
run PG_LEAN=1 PG_LEAN_INST_BLOCKS=6
run PG_LEAN=1 PG_FILL2_HIGH=1 PG_LEAN_INST_BLOCKS=5
run PG_LEAN=1
